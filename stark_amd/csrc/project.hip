// project.hip — per-element PSD projection (symx project_to_PD.cpp:12-32; ElementHessians.cpp:48-67,79-182) and the progressive-projection rounds:
// selection, eigen-decomposition kernels (LDS / register / translation-invariant variants), ordered matrix update, rounds started beside a solve.
#include "kernels_common.hpp"

namespace mistark {

// ======================================================================================================================
// PSD projection (project_to_PD.cpp:12-32; ElementHessians.cpp:48-67,79-182).
//   k_project_select_multi : marks the not-yet-projected elements that touch an active block row and appends them to a list
//   k_project_eig    : one WAVEFRONT per listed element: parallel-order cyclic Jacobi on the n x n matrix held in LDS
//                      (n/2 disjoint rotations per round, lanes own matrix entries), eigenvalues < eps clamped (or mirrored),
//                      V L V^T rebuilt only if something changed; the difference (projected - original) is added to the
//                      already assembled float BSR (the reference's update_global, ElementHessians.cpp:258-294).
// ======================================================================================================================

// selection for ALL potentials in one launch (one launch per potential was two dozen launches of a few microseconds each, 3.6 rounds per
// Newton iteration on configs[3]): marks the not-yet-projected elements that touch an active block row and appends them to their lists
struct SelDesc
{
    const int32_t* conn;
    const uint32_t* elem_list;   // sharded: the rank's elements (pools and keys are indexed by the position in this list)
    const int32_t* lrow;
    uint8_t* is_projected;
    uint32_t* list;              // selected elements as pool / key indices
    uint32_t* list_e;            // ... and as element numbers (what a lazy potential recomputes)
    int conn_stride, e_count, NB, counter, first_block, n_own;
    int dof_col[MAX_NB], dof_row_off[MAX_NB];
};
__global__ __launch_bounds__(BLOCK) void k_project_select_multi(const SelDesc* __restrict__ D, int n_desc, const uint8_t* __restrict__ active_blocks, int64_t* __restrict__ counters)
{
    __shared__ int s_k;
    if (threadIdx.x == 0) {
        int k = 0;
        while (k + 1 < n_desc && (int)blockIdx.x >= D[k + 1].first_block) k++;
        s_k = k;
    }
    __syncthreads();
    const SelDesc& d = D[s_k];
    const int le = ((int)blockIdx.x - d.first_block) * BLOCK + threadIdx.x;
    if (le >= d.e_count) return;
    const int e = d.elem_list ? (int)d.elem_list[le] : le;
    if (d.is_projected[le]) return;
    const int32_t* ce = d.conn + (size_t)e * d.conn_stride;
    if (active_blocks) {
        bool touch = false;
        for (int k = 0; k < d.NB; k++) touch = touch || active_blocks[d.dof_row_off[k] + ce[d.dof_col[k]]];
        if (!touch) return;
    }
    d.is_projected[le] = 1;
    // the selected lanes of a wavefront (all of one potential: a workgroup belongs to one descriptor) append with ONE atomic
    const unsigned long long sel = __ballot(1);
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)sel) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd((unsigned long long*)&counters[d.counter], (unsigned long long)__popcll(sel));
    base = ((unsigned long long)(unsigned int)__shfl((int)(base >> 32), leader, 64) << 32) | (unsigned int)__shfl((int)base, leader, 64);
    const unsigned long long idx = base + (unsigned long long)__popcll(sel & ((1ull << lane) - 1ull));
    d.list[idx] = (uint32_t)le;
    d.list_e[idx] = (uint32_t)e;
    // statistics: an element counts once, on the rank its energy counts on (energy_here)
    bool mine = true;
    if (d.lrow) {
        const int l = d.lrow[d.dof_row_off[0] + ce[d.dof_col[0]]];
        mine = l >= 0 && l < d.n_own;
    }
    const unsigned long long m = __ballot(mine);
    if (lane == leader && m) atomicAdd((unsigned long long*)&counters[3], (unsigned long long)__popcll(m));
}

// Pool addressing of the projection kernels: element e = list[li]; its blocks sit at H[(a*NB+b) * n_pool + pe], pe = e, or pe = li for a
// compact pool (the recomputed double blocks of a lazy potential's selected elements); slot_of_src is indexed by key: (a*NB+b) * n_elem + e.
template <int NB>
__global__ __launch_bounds__(BLOCK) void k_project_eig(double* __restrict__ elemH, int n_elem, int n_pool, int compact, const uint32_t* __restrict__ list, int n_list, double eps,
                                                       int mirroring, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals, int64_t* __restrict__ counters)
{
    constexpr int n = 3 * NB, nn = n * n, m = (n + 1) & ~1;  // m: even number of players of the round-robin schedule
    __shared__ double sA[4][nn], sV[4][nn], sC[4][m], sS[4][m], sL[4][m];
    __shared__ int sP[4][m];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave;
    if (w >= n_list) return;
    const int e = (int)list[w];
    const int pe = compact ? w : e;
    double* A = sA[wave];
    double* V = sV[wave];
    const size_t hs = (size_t)n_pool * 9;
    // load (block layout [a*NB+b][e][3][3]) and symmetrise exactly as stored
    for (int t = lane; t < nn; t += 64) {
        const int i = t / n, j = t - i * n;
        const int ba = i / 3, ii = i - 3 * ba, bb = j / 3, jj = j - 3 * bb;
        A[t] = elemH[(size_t)(ba * NB + bb) * hs + (size_t)pe * 9 + ii * 3 + jj];
        V[t] = i == j ? 1.0 : 0.0;
    }
    double fro = 0.0;
    for (int t = lane; t < nn; t += 64) fro += A[t] * A[t];
    fro = wave_sum(fro);
    fro = read_lane(fro, 0);
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
        for (int t = lane; t < nn; t += 64) {
            const int i = t / n, j = t - i * n;
            if (i != j) off += A[t] * A[t];
        }
        off = wave_sum(off);
        off = read_lane(off, 0);
        if (off <= JACOBI_OFF_TOL * fro) break;
        for (int r = 0; r < m - 1; r++) {
            // ---- rotations of this round: lane k < m/2 owns the pair (p, q)
            if (lane < m) {
                sC[wave][lane] = 1.0;
                sS[wave][lane] = 0.0;
                sP[wave][lane] = lane;
            }
            if (lane < m / 2) {
                int p, q;
                if (lane == 0) {
                    p = m - 1;
                    q = r;
                } else {
                    p = (r + lane) % (m - 1);
                    q = (r - lane + (m - 1)) % (m - 1);
                }
                if (p > q) {
                    const int tmp = p;
                    p = q;
                    q = tmp;
                }
                if (q < n) {
                    const double apq = A[p * n + q];
                    if (fabs(apq) > 1e-300) {
                        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                        // new[p] = cs old[p] - sn old[q];  new[q] = sn old[p] + cs old[q]
                        sC[wave][p] = cs;
                        sS[wave][p] = -sn;
                        sP[wave][p] = q;
                        sC[wave][q] = cs;
                        sS[wave][q] = sn;
                        sP[wave][q] = p;
                    }
                }
            }
            // (all LDS operations of one wavefront are performed in program order: no barrier needed inside the wave)
            // ---- rows: A <- J^T A
            {
                double v[(nn + 63) / 64];
                int c = 0;
                for (int t = lane; t < nn; t += 64, c++) {
                    const int i = t / n, j = t - i * n;
                    v[c] = sC[wave][i] * A[t] + sS[wave][i] * A[sP[wave][i] * n + j];
                }
                c = 0;
                for (int t = lane; t < nn; t += 64, c++) A[t] = v[c];
            }
            // ---- columns: A <- A J,  V <- V J
            {
                double v[(nn + 63) / 64], u[(nn + 63) / 64];
                int c = 0;
                for (int t = lane; t < nn; t += 64, c++) {
                    const int i = t / n, j = t - i * n;
                    const int pj = sP[wave][j];
                    v[c] = sC[wave][j] * A[t] + sS[wave][j] * A[i * n + pj];
                    u[c] = sC[wave][j] * V[t] + sS[wave][j] * V[i * n + pj];
                }
                c = 0;
                for (int t = lane; t < nn; t += 64, c++) {
                    A[t] = v[c];
                    V[t] = u[c];
                }
            }
        }
    }
    // eigenvalues = diag(A)
    bool bad = false;
    if (lane < n) {
        double l = A[lane * n + lane];
        if (l < eps) {
            bad = true;
            l = (mirroring & 1) ? -l : eps;  // (bit 1: k_project_eig_cols' IEEE switch)
        }
        sL[wave][lane] = l;
    }
    const bool changed = __ballot(bad) != 0ull;
    if (lane == 0 && changed) atomicAdd((unsigned long long*)&counters[1], 1ull);
    if (!changed) return;  // untouched, like the reference (project_to_PD.cpp:25-29)
    for (int t = lane; t < nn; t += 64) {
        const int i = t / n, j = t - i * n;
        double acc = 0.0;
        for (int k = 0; k < n; k++) acc += V[i * n + k] * sL[wave][k] * V[j * n + k];
        const int ba = i / 3, ii = i - 3 * ba, bb = j / 3, jj = j - 3 * bb;
        const size_t blk = (size_t)(ba * NB + bb) * n_elem + e;
        double* dst = elemH + (size_t)(ba * NB + bb) * hs + (size_t)pe * 9 + ii * 3 + jj;
        if (vals) {
            const uint32_t slot = slot_of_src[blk];
            if (slot != NO_SRC) atomicAdd(&vals[tile_val_index(slot, ii * 3 + jj)], (float)(acc - *dst));  // (NO_SRC: the block row belongs to another rank)
        }
        *dst = acc;
    }
}

// ---- the same projection with the matrix in REGISTERS ---------------------------------------------------------------------------------
// k_project_eig keeps A and V in LDS and is bound by the LDS pipe (≈ 45 64-lane, 8-byte LDS operations per rotation round and element:
// 30 ns per 12 x 12 element on the whole chip). Here one group of m = even(n) lanes owns an element, lane c holds COLUMN c of A and of V
// in registers, 64 / m elements share a wavefront. One round:
//   every lane fetches its partner's two columns with ds_bpermute (no bank storage involved), both lanes of a pair compute the same
//   rotation from the same three numbers and update their columns (A J, V J); the row rotations J^T need, in every column, the entry of
//   the partner ROW: the updated columns go through LDS once (row-major, conflict-free) and come back as y[partner(i)].
// ≈ 1/4 of the LDS bytes per element and round. Same cyclic-by-round Jacobi, same pair schedule, same threshold and clamping as
// k_project_eig; an element's result does not depend on which other elements share its wavefront (a converged group applies identity
// rotations).
template <int n>
__device__ __forceinline__ double pick(const double (&x)[n], int idx)
{
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < n; i++) r = i == idx ? x[i] : r;
    return r;
}
// partner of player i in round r of the round-robin schedule of m players (k_project_eig: pairs (m-1, r), ((r+k) % (m-1), (r-k) % (m-1)))
__device__ __forceinline__ int rr_partner(int m, int r, int i) { return i == m - 1 ? r : (i == r ? m - 1 : (2 * r - i + 2 * (m - 1)) % (m - 1)); }
template <int NB>
struct ProjWaveShared  // LDS of ONE wavefront (waves of a block may work on different potentials, even different NB)
{
    static constexpr int n = 3 * NB, m = (n + 1) & ~1, W = m, EPW = 64 / W;
    double M[EPW + 1][m * W];   // [row][column] of the element of a group: row exchange; eigenvectors for the rebuild (+1: idle tail lanes)
    double2 CS[EPW + 1][m];     // (c, s) of the current round by player
    double L[EPW + 1][m];       // clamped eigenvalues
    double R[64 + W];           // group sums
};
// w = index of this wavefront within the list (EPW elements each)
template <int NB>
__device__ __forceinline__ void project_cols_body(ProjWaveShared<NB>& S, int w, double* __restrict__ elemH, int n_elem, int n_pool, int compact, const uint32_t* __restrict__ list,
                                                  int n_list, double eps, int mirroring, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals,
                                                  int64_t* __restrict__ counters)
{
    constexpr int n = 3 * NB, nn = n * n, m = (n + 1) & ~1, W = m, EPW = 64 / W;
    const int lane = threadIdx.x & 63;
    const int g = lane / W, c = lane - g * W;
    if (w * EPW >= n_list) return;  // (whole wavefront)
    const int li = w * EPW + g;
    const bool elem_ok = g < EPW && li < n_list;
    const bool valid = elem_ok && c < n;  // this lane holds a column
    const int e = elem_ok ? (int)list[li] : 0;
    const int pe = compact ? (elem_ok ? li : 0) : e;
    const size_t hs = (size_t)n_pool * 9;
    const int bb = c / 3, jj = c - 3 * bb;
    double* M = S.M[g];
    double a[n], v[n];
#pragma unroll
    for (int i = 0; i < n; i++) {
        const int ba = i / 3, ii = i - 3 * ba;
        a[i] = valid ? elemH[(size_t)(ba * NB + bb) * hs + (size_t)pe * 9 + ii * 3 + jj] : 0.0;
        v[i] = i == c ? 1.0 : 0.0;
    }
    auto group_sum = [&](double x) {
        S.R[lane] = x;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) sum += S.R[g * W + k];
        return sum;
    };
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < n; i++) fro += a[i] * a[i];
    fro = group_sum(fro);
    bool active = elem_ok;
    // 1 / sqrt(x) to double precision from the hardware estimate and two Newton steps (a rotation only has to be orthogonal to rounding,
    // c^2 + s^2 = 1; its angle may be a few ulps off the ideal one: that costs nothing, the sweeps iterate anyway)
    auto rsqrt_nr = [](double x) {
        double y = __builtin_amdgcn_rsq(x);
        y = y * (1.5 - 0.5 * x * y * y);
        return y * (1.5 - 0.5 * x * y * y);
    };
    auto rcp_nr = [](double x) {
        const double y = __builtin_amdgcn_rcp(x);
        return fma(y, fma(-x, y, 1.0), y);
    };
    auto shfl64 = [](double x, int addr) {  // addr = 4 * source lane
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
        return __hiloint2double(hi, lo);
    };
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++) off += i == c ? 0.0 : a[i] * a[i];
        off = group_sum(off);
        if (off <= JACOBI_OFF_TOL * fro) active = false;
        if (__ballot(active) == 0ull) break;
        // (rounds unrolled: the row rotations then address the lane's own registers with constant indices, see project_ti_body)
#pragma unroll
        for (int r = 0; r < m - 1; r++) {
            const int partner = rr_partner(m, r, c);
            const int src = ((g * W + partner) & 63) << 2;
            const bool is_lo = c < partner;
            // the three numbers of my pair's rotation: A[lo][lo], A[hi][hi], A[hi][lo] (the entry the lower lane holds)
            const double d_own = pick<n>(a, c), x_own = pick<n>(a, partner);
            const double d_oth = shfl64(d_own, src), x_oth = shfl64(x_own, src);
            double cs = 1.0, sg = 0.0;  // my column <- cs * mine + sg * partner's
            if (active && c < n && partner < n) {
                const double app = is_lo ? d_own : d_oth, aqq = is_lo ? d_oth : d_own, apq = is_lo ? x_own : x_oth;
                if (fabs(apq) > 1e-300) {
                    // tan of the rotation angle from the hardware reciprocal / reciprocal-square-root estimates + one Newton step each
                    // (the IEEE division and square root sequences, with their scaling and fix-up code, were the longest dependent chain
                    // of a round; a rotation only has to be orthogonal to rounding, which cs below takes care of)
                    if (mirroring & 2) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        cs = 1.0 / sqrt(t * t + 1.0);
                        const double sn = t * cs;
                        sg = is_lo ? -sn : sn;
                    } else {
                    const double theta = (aqq - app) * rcp_nr(2.0 * apq);
                    const double s2 = fma(theta, theta, 1.0);
                    double y = __builtin_amdgcn_rsq(s2);
                    y = y * (1.5 - 0.5 * s2 * y * y);
                    const double root = s2 < 1e300 ? s2 * y : fabs(theta);
                    const double t = copysign(rcp_nr(fabs(theta) + root), theta);
                    cs = rsqrt_nr(t * t + 1.0);
                    const double sn = t * cs;
                    sg = is_lo ? -sn : sn;  // new[lo] = cs old[lo] - sn old[hi];  new[hi] = sn old[lo] + cs old[hi]
                    }
                }
            }
            S.CS[g][c] = make_double2(cs, sg);
            // columns: A <- A J, V <- V J
#pragma unroll
            for (int i = 0; i < n; i++) {
                a[i] = cs * a[i] + sg * shfl64(a[i], src);
                v[i] = cs * v[i] + sg * shfl64(v[i], src);
            }
            // rows: A <- J^T A, pair by pair
#pragma unroll
            for (int i = 0; i < n; i++) {
                const int pi = rr_partner(m, r, i);
                if (pi > i && pi < n) {
                    const double2 ri = S.CS[g][i], rp = S.CS[g][pi];
                    const double ai = a[i], ap_ = a[pi];
                    a[i] = ri.x * ai + ri.y * ap_;
                    a[pi] = rp.x * ap_ + rp.y * ai;
                }
            }
        }
    }
    // eigenvalues = diag(A)
    double l = pick<n>(a, c);
    bool bad = false;
    if (valid && l < eps) {
        bad = true;
        l = (mirroring & 1) ? -l : eps;
    }
    const unsigned long long bad_mask = __ballot(bad);
    const bool changed = elem_ok && ((bad_mask >> (g * W)) & ((1ull << W) - 1ull)) != 0ull;
    if (changed && c == 0) atomicAdd((unsigned long long*)&counters[1], 1ull);
    if (__ballot(changed) == 0ull) return;  // untouched, like the reference (project_to_PD.cpp:25-29)
#pragma unroll
    for (int i = 0; i < n; i++) M[i * W + c] = v[i];
    S.L[g][c] = l;
    if (!(changed && valid)) return;
    double wc[n];  // row c of V
#pragma unroll
    for (int k = 0; k < n; k++) wc[k] = M[c * W + k];
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < n; k++) acc = fma(M[i * W + k] * wc[k], S.L[g][k], acc);  // (V_ik V_ck) l_k: symmetric in (i, c) to the bit
        const int ba = i / 3, ii = i - 3 * ba;
        const size_t blk = (size_t)(ba * NB + bb) * n_elem + e;
        double* dst = elemH + (size_t)(ba * NB + bb) * hs + (size_t)pe * 9 + ii * 3 + jj;
        if (vals) {
            const uint32_t slot = slot_of_src[blk];
            if (slot != NO_SRC) atomicAdd(&vals[tile_val_index(slot, ii * 3 + jj)], (float)(acc - *dst));  // (NO_SRC: the block row belongs to another rank)
        }
        *dst = acc;
    }
}

// ---- the same projection for TRANSLATION-INVARIANT elements (tets, membrane triangles: the energy depends on differences of node positions
// only), on a matrix of 3 (NB - 1) instead of 3 NB rows. Such an element Hessian H annihilates the three rigid translations exactly, so in
// the node basis Q = [q_0 .. q_{NB-2} | 1/sqrt(NB)] (Helmert: orthonormal, the first NB - 1 columns sum to zero) it reads
//     (Q x I3)^T H (Q x I3) = [ A'  0 ; 0  0 ],      A' = 3 (NB - 1) square,
// its eigenvalues are those of A' plus three zeros, and its eigenvectors the back-transformed ones of A' plus the translations. The
// reference's dense eigen-solver finds the three zeros as +-1e-16 ||H|| and clamps them like any eigenvalue below eps (to eps, or to their
// mirror image): here they are clamped as exact zeros. What is saved: Jacobi on 9 x 9 instead of 12 x 12 costs (9/12)^3 of the rotations,
// 9 instead of 11 rounds per sweep, and six elements share a wavefront instead of five (membrane triangles: 6 x 6 instead of 9 x 9).
// Differences to the full-size path: <= eps in the null directions (the clamped value of a numerical zero), rounding elsewhere; every
// element counts as changed (it always has three eigenvalues below eps), which is what the reference reports for them, too.
template <int NB>
struct ProjTiShared  // LDS of ONE wavefront
{
    static constexpr int n = 3 * (NB - 1), m = (n + 1) & ~1, W = m, EPW = 64 / W;
    double M[EPW + 1][m * W];   // row exchange during the sweeps, then the eigenvectors
    double P[EPW + 1][m * W];   // the rebuilt reduced matrix
    double2 CS[EPW + 1][m];
    double L[EPW + 1][m];
    double R[64 + W];
};
// Helmert basis of NB nodes: column k < NB - 1 (column NB - 1 is the constant 1 / sqrt(NB))
__device__ __forceinline__ double helmert(int i, int k)
{
    const double s = rsqrt((double)((k + 1) * (k + 2)));
    return i <= k ? s : (i == k + 1 ? -(double)(k + 1) * s : 0.0);
}
template <int NB>
__device__ __forceinline__ void project_ti_body(ProjTiShared<NB>& S, int w, double* __restrict__ elemH, int n_elem, int n_pool, int compact, const uint32_t* __restrict__ list, int n_list,
                                                double eps, int mirroring, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals, int64_t* __restrict__ counters)
{
    constexpr int n = 3 * (NB - 1), m = (n + 1) & ~1, W = m, EPW = 64 / W;
    const int lane = threadIdx.x & 63;
    const int g = lane / W, c = lane - g * W;
    if (w * EPW >= n_list) return;  // (whole wavefront)
    const int li = w * EPW + g;
    const bool elem_ok = g < EPW && li < n_list;
    const bool valid = elem_ok && c < n;  // this lane holds a column of the reduced matrix
    const int e = elem_ok ? (int)list[li] : 0;
    const int pe = compact ? (elem_ok ? li : 0) : e;
    const size_t hs = (size_t)n_pool * 9;
    double* M = S.M[g];
    double* Pm = S.P[g];
    double a[n], v[n];
    {
        // column c = (node-basis vector ap, component cc) of A' = (Q x I)^T H (Q x I), formed while loading
        const int ap = valid ? c / 3 : 0, cc = valid ? c - 3 * (c / 3) : 0;
        double qa[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) qa[i] = helmert(i, ap);
        double T[NB][3];
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int ci = 0; ci < 3; ci++) {
                double t = 0.0;
#pragma unroll
                for (int b = 0; b < NB; b++) t += qa[b] * (valid ? elemH[(size_t)(i * NB + b) * hs + (size_t)pe * 9 + ci * 3 + cc] : 0.0);
                T[i][ci] = t;
            }
#pragma unroll
        for (int ip = 0; ip < NB - 1; ip++)
#pragma unroll
            for (int ci = 0; ci < 3; ci++) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < NB; i++) t += helmert(i, ip) * T[i][ci];
                a[ip * 3 + ci] = t;
            }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = i == c ? 1.0 : 0.0;
    }
    auto group_sum = [&](double x) {
        S.R[lane] = x;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) sum += S.R[g * W + k];
        return sum;
    };
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < n; i++) fro += a[i] * a[i];
    fro = group_sum(fro);
    bool active = elem_ok;
    auto rsqrt_nr = [](double x) {
        double y = __builtin_amdgcn_rsq(x);
        y = y * (1.5 - 0.5 * x * y * y);
        return y * (1.5 - 0.5 * x * y * y);
    };
    auto rcp_nr = [](double x) {
        const double y = __builtin_amdgcn_rcp(x);
        return fma(y, fma(-x, y, 1.0), y);
    };
    auto shfl64 = [](double x, int addr) {  // addr = 4 * source lane
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
        return __hiloint2double(hi, lo);
    };
    // The sweeps: as in project_cols_body, but with the rounds of a sweep unrolled. The row rotations of a round touch, in every column, the
    // two entries of each rotated pair — both in the lane's own registers; with the round index a compile-time constant so are their
    // indices, and the column no longer travels through LDS to be indexed at run time (a write, a read and the index arithmetic per entry
    // and round). Only the rotations themselves (c, s per column) still go through LDS.
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++) off += i == c ? 0.0 : a[i] * a[i];
        off = group_sum(off);
        if (off <= JACOBI_OFF_TOL * fro) active = false;
        if (__ballot(active) == 0ull) break;
#pragma unroll
        for (int r = 0; r < m - 1; r++) {
            const int partner = rr_partner(m, r, c);
            const int src = ((g * W + partner) & 63) << 2;
            const bool is_lo = c < partner;
            const double d_own = pick<n>(a, c), x_own = pick<n>(a, partner);
            const double d_oth = shfl64(d_own, src), x_oth = shfl64(x_own, src);
            double cs = 1.0, sg = 0.0;
            if (active && c < n && partner < n) {
                const double app = is_lo ? d_own : d_oth, aqq = is_lo ? d_oth : d_own, apq = is_lo ? x_own : x_oth;
                if (fabs(apq) > 1e-300) {
                    if (mirroring & 2) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        cs = 1.0 / sqrt(t * t + 1.0);
                        const double sn = t * cs;
                        sg = is_lo ? -sn : sn;
                    } else {
                        const double theta = (aqq - app) * rcp_nr(2.0 * apq);
                        const double s2 = fma(theta, theta, 1.0);
                        double y = __builtin_amdgcn_rsq(s2);
                        y = y * (1.5 - 0.5 * s2 * y * y);
                        const double root = s2 < 1e300 ? s2 * y : fabs(theta);
                        const double t = copysign(rcp_nr(fabs(theta) + root), theta);
                        cs = rsqrt_nr(t * t + 1.0);
                        const double sn = t * cs;
                        sg = is_lo ? -sn : sn;
                    }
                }
            }
            S.CS[g][c] = make_double2(cs, sg);
            // columns: A <- A J, V <- V J
#pragma unroll
            for (int i = 0; i < n; i++) {
                a[i] = cs * a[i] + sg * shfl64(a[i], src);
                v[i] = cs * v[i] + sg * shfl64(v[i], src);
            }
            // rows: A <- J^T A, pair by pair (indices are constants after unrolling)
#pragma unroll
            for (int i = 0; i < n; i++) {
                const int pi = rr_partner(m, r, i);
                if (pi > i && pi < n) {
                    const double2 ri = S.CS[g][i], rp = S.CS[g][pi];
                    const double ai = a[i], ap_ = a[pi];
                    a[i] = ri.x * ai + ri.y * ap_;
                    a[pi] = rp.x * ap_ + rp.y * ai;
                }
            }
        }
    }
    double l = pick<n>(a, c);
    if (valid && l < eps) l = (mirroring & 1) ? -l : eps;
    const double null_val = (mirroring & 1) ? 0.0 : eps;  // what the three exact zeros become
    if (elem_ok && c == 0) atomicAdd((unsigned long long*)&counters[1], 1ull);
    // the rebuilt reduced matrix P = V diag(l) V^T (column c in this lane, symmetric in (i, c) to the bit), through LDS
#pragma unroll
    for (int i = 0; i < n; i++) M[i * W + c] = v[i];
    S.L[g][c] = l;
    if (valid) {
        double wc[n];
#pragma unroll
        for (int k = 0; k < n; k++) wc[k] = M[c * W + k];
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) acc = fma(M[i * W + k] * wc[k], S.L[g][k], acc);
            Pm[i * W + c] = acc;
        }
    }
    if (!elem_ok) return;
    // back to the node basis: H' = (Q x I) P (Q x I)^T + null_val / NB on the (ci == cc) entries of every block. Output column oc = (node a,
    // component cc); every unordered pair of entries is computed once, by the lane of the smaller index, and written to both places.
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const int oc = pass == 0 ? c : n + c;
        if (pass == 0 ? c >= n : c >= 3) continue;
        const int ao = oc / 3, cc = oc - 3 * ao;
        double U[NB - 1][3];  // sum over the column's node-basis index
#pragma unroll
        for (int ip = 0; ip < NB - 1; ip++)
#pragma unroll
            for (int ci = 0; ci < 3; ci++) {
                double t = 0.0;
#pragma unroll
                for (int bp = 0; bp < NB - 1; bp++) t += helmert(ao, bp) * Pm[(ip * 3 + ci) * W + (bp * 3 + cc)];
                U[ip][ci] = t;
            }
#pragma unroll 1
        for (int i = 0; i < NB; i++) {
#pragma unroll
            for (int ci = 0; ci < 3; ci++) {
                const int orow = i * 3 + ci;
                if (orow < oc) continue;  // (computed by the lane of column orow)
                double acc = ci == cc ? null_val / (double)NB : 0.0;
#pragma unroll
                for (int ip = 0; ip < NB - 1; ip++) acc += helmert(i, ip) * U[ip][ci];
                // entry (row (i, ci), column (ao, cc)) and its mirror image
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (side == 1 && orow == oc) break;
                    const int bi = side == 0 ? i : ao, bj = side == 0 ? ao : i, ii = side == 0 ? ci : cc, jj = side == 0 ? cc : ci;
                    double* dst = elemH + (size_t)(bi * NB + bj) * hs + (size_t)pe * 9 + ii * 3 + jj;
                    if (vals) {
                        const uint32_t slot = slot_of_src[(size_t)(bi * NB + bj) * n_elem + e];
                        if (slot != NO_SRC) atomicAdd(&vals[tile_val_index(slot, ii * 3 + jj)], (float)(acc - *dst));
                    }
                    *dst = acc;
                }
            }
        }
    }
}
// (one wavefront per workgroup: a wavefront needs 13.5 KB of LDS, and 160 KB hold eleven single-wavefront workgroups but only two of four)
template <int NB>
__global__ __launch_bounds__(64) void k_project_eig_ti(double* __restrict__ elemH, int n_elem, int n_pool, int compact, const uint32_t* __restrict__ list, int n_list, double eps,
                                                       int mirroring, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals, int64_t* __restrict__ counters)
{
    __shared__ ProjTiShared<NB> S;
    project_ti_body<NB>(S, blockIdx.x, elemH, n_elem, n_pool, compact, list, n_list, eps, mirroring, slot_of_src, vals, counters);
}

template <int NB>
__global__ __launch_bounds__(BLOCK) void k_project_eig_cols(double* __restrict__ elemH, int n_elem, int n_pool, int compact, const uint32_t* __restrict__ list, int n_list, double eps,
                                                            int mirroring, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals, int64_t* __restrict__ counters)
{
    __shared__ ProjWaveShared<NB> S[4];
    const int wave = threadIdx.x >> 6;
    project_cols_body<NB>(S[wave], blockIdx.x * 4 + wave, elemH, n_elem, n_pool, compact, list, n_list, eps, mirroring, slot_of_src, vals, counters);
}
// The short lists of one projection round (contact kinds with a few dozen rows, the rigid-body potentials, ...) in ONE launch: a lone
// wavefront needs 100-300 us for its elements whatever their number (≈ 90 dependent rotation rounds), a dozen such launches in a row is
// where the time of a round went. Every wavefront looks up the potential it works for.
struct ProjDesc
{
    double* H;
    const uint32_t* list;
    const uint32_t* sos;
    float* vals;
    int n_elem, nl, NB, first_wave;
    int n_pool, compact;
    int ti;  // translation-invariant elements: reduced matrix (project_ti_body)
};
constexpr int PROJ_BATCH = 40;
struct ProjBatch
{
    ProjDesc d[PROJ_BATCH];
    int n;
};
union ProjWaveSharedAny
{
    ProjWaveShared<1> s1;
    ProjWaveShared<2> s2;
    ProjWaveShared<3> s3;
    ProjWaveShared<4> s4;
    ProjWaveShared<5> s5;
    ProjWaveShared<6> s6;
    ProjTiShared<3> t3;
    ProjTiShared<4> t4;
    __device__ ProjWaveSharedAny() {}
};
__global__ __launch_bounds__(BLOCK) void k_project_eig_multi(ProjBatch B, double eps, int mirroring, int64_t* __restrict__ counters)
{
    __shared__ ProjWaveSharedAny S[4];
    const int wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    int k = 0;
    while (k + 1 < B.n && gw >= B.d[k + 1].first_wave) k++;
    const ProjDesc& D = B.d[k];
    const int w = gw - D.first_wave;
    if (D.ti) {
        if (D.NB == 4) project_ti_body<4>(S[wave].t4, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters);
        else project_ti_body<3>(S[wave].t3, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters);
        return;
    }
    switch (D.NB) {
        case 1: project_cols_body<1>(S[wave].s1, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
        case 2: project_cols_body<2>(S[wave].s2, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
        case 3: project_cols_body<3>(S[wave].s3, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
        case 4: project_cols_body<4>(S[wave].s4, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
        case 5: project_cols_body<5>(S[wave].s5, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
        default: project_cols_body<6>(S[wave].s6, w, D.H, D.n_elem, D.n_pool, D.compact, D.list, D.nl, eps, mirroring, D.sos, D.vals, counters); break;
    }
}

// (sharded: the gradient is complete on the rank's rows and, after the halo exchange, its ghosts; other rows hold partial sums nobody reads.
// Inactive rows are counted over the rank's own rows.)
__global__ __launch_bounds__(BLOCK) void k_active_blocks(const double* __restrict__ grad, int64_t nbr, double thr, uint8_t* __restrict__ active, int64_t* __restrict__ counters,
                                                        const int32_t* __restrict__ lrow, int n_own)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in = r < nbr;
    const double m = in ? fmax(fabs(grad[3 * r]), fmax(fabs(grad[3 * r + 1]), fabs(grad[3 * r + 2]))) : 0.0;
    const bool act = !in || m >= thr;
    if (in) active[r] = act ? 1 : 0;
    const bool own = in && (!lrow || (lrow[r] >= 0 && lrow[r] < n_own));
    // (atomics on ONE address serialise at ~10 ns each: one per row took 34 us, one per wavefront still 30; one per workgroup)
    __shared__ int s_cnt[BLOCK / 64];
    const unsigned long long inactive = __ballot(!act && own);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(inactive);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (n) atomicAdd((unsigned long long*)&counters[2], (unsigned long long)n);
    }
}

void project_spec_discard(Context& c);
// Ordered update of the assembled matrix after a projection round (instead of float deltas added atomically in arrival order): every block
// a selected element contributes to is flagged, and the flagged blocks are gathered again from the pools in sorted-key order (gather_part) —
// the matrix equals the one assembled from the projected Hessians, bit for bit and run to run. For a lazy potential (float upper-triangle
// pool, blocks recomputed into a compact double pool for the projection) the projected blocks go back to the float pool first; a block
// whose floats did not change flags nothing. One thread per (selected element, block pair).
struct MarkDesc
{
    const uint32_t* list;
    const uint32_t* slot_of_src;
    uint8_t* dirty;
    const double* Hc;
    float* hf;
    int nl, NB, n_key, n_pool_c, n_pool_f;
    int first_block;  // of this list in the common grid
    int hf_element_major;
};
constexpr int MARK_BATCH = 16;
struct MarkBatch  // every potential's list of one projection round in ONE launch (nine launches of 4.6 us each on configs[3])
{
    MarkDesc d[MARK_BATCH];
    int n;
};
__global__ __launch_bounds__(BLOCK) void k_proj_mark(MarkBatch mb)
{
    int k = 0;
    while (k + 1 < mb.n && (int)blockIdx.x >= mb.d[k + 1].first_block) k++;
    const MarkDesc& D = mb.d[k];
    const uint32_t* __restrict__ list = D.list;
    const uint32_t* __restrict__ slot_of_src = D.slot_of_src;
    uint8_t* __restrict__ dirty = D.dirty;
    const double* __restrict__ Hc = D.Hc;
    float* __restrict__ hf = D.hf;
    const int nl = D.nl, NB = D.NB, n_key = D.n_key, n_pool_c = D.n_pool_c, n_pool_f = D.n_pool_f;
    const int64_t t = (int64_t)((int)blockIdx.x - D.first_block) * BLOCK + threadIdx.x;
    const int nn = NB * NB;
    if (t >= (int64_t)nl * nn) return;
    const int li = (int)(t / nn), ab = (int)(t - (int64_t)li * nn), a = ab / NB, b = ab - a * NB;
    const uint32_t le = list[li];
    const uint32_t slot = slot_of_src[(size_t)ab * n_key + le];
    if (!hf) {
        if (slot != NO_SRC) dirty[slot] = 1;
        return;
    }
    if (a > b) return;  // (the pool holds the upper block triangle; (b, a) is read as the transpose of (a, b))
    const double* src = Hc + ((size_t)ab * n_pool_c + li) * 9;
    float* dst = hf + (D.hf_element_major ? (size_t)le * 10 + tet_pair_index(a, b) : (size_t)tet_pair_index(a, b) * n_pool_f + le) * 9;
    bool diff = false;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float f = (float)src[k];
        if (dst[k] != f) {
            dst[k] = f;
            diff = true;
        }
    }
    if (!diff) return;
    if (slot != NO_SRC) dirty[slot] = 1;
    if (a != b) {
        const uint32_t slot_t = slot_of_src[(size_t)(b * NB + a) * n_key + le];
        if (slot_t != NO_SRC) dirty[slot_t] = 1;
    }
}

// project() in three phases, so that a round can be started AHEAD of the solve that may need it (project_speculate below):
//   A  selection: which rows are active (by the gradient), which elements touch them, the lists per potential — ends with the counts on the device
//   B  (needs the counts on the host) the eigen-projections of the listed elements, in their pools; nothing of the assembled matrix is touched
//      when the update is ordered (marks)
//   C  the marks: projected blocks back into the float pool, the touched matrix blocks gathered again in sorted-key order; statistics
static void project_phase_a(Context& c, const uint8_t* active_host, bool by_gradient, double threshold)
{
    const int np = (int)c.pots.size();
    if (np + 4 > 128) throw Error("project: too many potentials");
    c.counters.ensure(128);
    fill_async(c.stream, c.counters.p, 0, 128 * sizeof(int64_t));
    const uint8_t* act = nullptr;
    const int32_t* lrow = c.world > 1 ? c.sh.lrow.p : nullptr;
    if (by_gradient) {
        hipLaunchKernelGGL(k_active_blocks, dim3(grid_for(c.nbr)), dim3(BLOCK), 0, c.stream, c.grad.p, c.nbr, threshold, c.active_blocks.p, c.counters.p, lrow, (int)c.sh.n_own);
        act = c.active_blocks.p;
    } else if (active_host) {
        MS_CHECK(hipMemcpyAsync(c.active_blocks.p, active_host, (size_t)c.nbr, hipMemcpyHostToDevice, c.stream));
        act = c.active_blocks.p;
    }
    // selection: per-potential lists (counter 4 + potential index); counter 3: selected elements whose energy counts on this rank
    c.proj_list.ensure(2 * std::max<size_t>(c.n_elem_total, 1));
    uint32_t* list_e_base = c.proj_list.p + std::max<size_t>(c.n_elem_total, 1);
    // (the table lives in the context: the copy below may still read it after this scope; the read-back that follows the selection
    // orders it before the next round overwrites it)
    c.sel_desc_host.resize((size_t)np * sizeof(SelDesc));
    SelDesc* desc_h = reinterpret_cast<SelDesc*>(c.sel_desc_host.data());
    int n_desc = 0;
    int n_blocks = 0;
    for (int pi = 0; pi < np; pi++) {
        Potential& P = c.pots[pi];
        if (P.args.e_count == 0) continue;
        SelDesc d{};
        d.conn = P.args.conn;
        d.elem_list = P.args.elem_list;
        d.lrow = lrow;
        d.n_own = (int)c.sh.n_own;
        d.is_projected = c.is_projected.p + P.e_off;
        d.list = c.proj_list.p + P.e_off;
        d.list_e = list_e_base + P.e_off;
        d.conn_stride = P.args.conn_stride;
        d.e_count = P.args.e_count;
        d.NB = P.NB;
        d.counter = 4 + pi;
        d.first_block = n_blocks;
        for (int k = 0; k < MAX_NB; k++) {
            d.dof_col[k] = P.args.dof_col[k];
            d.dof_row_off[k] = P.args.dof_row_off[k];
        }
        n_blocks += (P.args.e_count + BLOCK - 1) / BLOCK;
        desc_h[n_desc++] = d;
    }
    if (n_desc > 0) {
        c.sel_desc.ensure((size_t)n_desc * sizeof(SelDesc));
        MS_CHECK(hipMemcpyAsync(c.sel_desc.p, desc_h, (size_t)n_desc * sizeof(SelDesc), hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(k_project_select_multi, dim3(n_blocks), dim3(BLOCK), 0, c.stream, (const SelDesc*)c.sel_desc.p, n_desc, act, c.counters.p);
    }
}
// h: the counters of phase A on the host. ordered: the eigen kernels leave the matrix alone and the touched blocks are gathered again in phase C
// (marks); otherwise they patch the matrix themselves where it is current (atomic deltas).
static void project_phase_b(Context& c, const int64_t* h, double eps, int mirroring, Context::ProjRound& R)
{
    const int np = (int)c.pots.size();
    uint32_t* list_e_base = c.proj_list.p + std::max<size_t>(c.n_elem_total, 1);
    R.marks.clear();
    R.mark_part[0] = R.mark_part[1] = false;
    // eigen-projection of the selected elements; deltas go straight into the assembled matrix if it is current (rows of other ranks:
    // their owners project the same element and get the same numbers)
    if (c.proj_variant & 4) mirroring |= 2;
    constexpr int SHORT_LIST = 4096;  // lists up to this length share one launch (k_project_eig_multi)
    ProjBatch batch;
    batch.n = 0;
    int batch_waves = 0;
    auto flush = [&]() {
        if (batch.n == 0) return;
        hipLaunchKernelGGL(k_project_eig_multi, dim3((batch_waves + 3) / 4), dim3(BLOCK), 0, c.stream, batch, eps, mirroring, c.counters.p);
        batch.n = 0;
        batch_waves = 0;
    };
    for (int pi = 0; pi < np; pi++) {
        Potential& P = c.pots[pi];
        const int nl = (int)h[4 + pi];
        if (nl == 0) continue;
        hipStream_t stream = c.stream;
        double* H = c.elemH.p + P.h_off;
        const uint32_t* list = c.proj_list.p + P.e_off;
        const uint32_t* sos = c.part[P.part].slot_of_src.p + P.kp_off;
        const int n_key = P.n_key;
        int n_pool = P.n_key, compact = 0;
        if (c.lazy_active && P.lazy_capable) {
            // the double blocks of the selected elements were never stored: recompute them into a compact pool (the list is a small
            // fraction of the mesh except when PPN activates every element, and then the eigen-decompositions cost 20x this)
            // The compact pool is the potential's own share of the double pool, which the lazy path leaves unused (prepare(): sized for whole
            // wavefronts). A separate buffer sized by the round was a 1.15 GB allocation INSIDE the Newton loop the first time PPN activated every
            // tet of configs[3] — 5 to 50 ms from box to box, up to a sixth of bench.py's timed window.
            n_pool = (nl + 63) / 64 * 64;
            H = c.elemH.p + P.h_off;
            compact = 1;
            launch_tet_closed_list(c, P, list_e_base + P.e_off, nl, H, n_pool);
        }
        // patched in place where the matrix already holds these Hessians: all of it after assemble(), its static part after eval()'s early gather
        float* vals = (c.matrix_current || (c.static_assembled && P.part == 0)) ? c.part[P.part].vals.p : nullptr;
        if (vals && !c.atomic_projection) {  // ordered update: the kernels leave the matrix alone, the touched blocks are gathered again in phase C
            R.marks.push_back(Context::ProjRound::Mark{pi, list, nl, compact ? H : (const double*)nullptr, n_pool});
            R.mark_part[P.part] = true;
            vals = nullptr;
        }
        const dim3 g((nl + 3) / 4), b(BLOCK);
        if (!(c.proj_variant & 1) && P.NB <= 6) {  // register-resident Jacobi, several elements per wavefront
            const int epw = 64 / ((3 * P.NB + 1) & ~1);
            const bool ti = P.ti_projection && !(c.proj_variant & 8) && (P.NB == 3 || P.NB == 4);
            if (nl <= SHORT_LIST && !(c.proj_variant & 2)) {
                if (batch.n == PROJ_BATCH) flush();
                const int epw_b = ti ? 64 / ((3 * (P.NB - 1) + 1) & ~1) : epw;
                batch.d[batch.n++] = ProjDesc{H, list, sos, vals, n_key, nl, P.NB, batch_waves, n_pool, compact, ti ? 1 : 0};
                batch_waves += (nl + epw_b - 1) / epw_b;
                continue;
            }
            if (ti) {  // translation-invariant elements: reduced matrix (k_project_eig_ti)
                const int epw_ti = 64 / ((3 * (P.NB - 1) + 1) & ~1);
                const dim3 grid_ti((nl + epw_ti - 1) / epw_ti), b_ti(64);
                if (P.NB == 4) hipLaunchKernelGGL((k_project_eig_ti<4>), grid_ti, b_ti, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p);
                else hipLaunchKernelGGL((k_project_eig_ti<3>), grid_ti, b_ti, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p);
                continue;
            }
            const dim3 grid(((nl + epw - 1) / epw + 3) / 4);
            switch (P.NB) {
                case 1: hipLaunchKernelGGL((k_project_eig_cols<1>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
                case 2: hipLaunchKernelGGL((k_project_eig_cols<2>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
                case 3: hipLaunchKernelGGL((k_project_eig_cols<3>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
                case 4: hipLaunchKernelGGL((k_project_eig_cols<4>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
                case 5: hipLaunchKernelGGL((k_project_eig_cols<5>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
                default: hipLaunchKernelGGL((k_project_eig_cols<6>), grid, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            }
            continue;
        }
        switch (P.NB) {
            case 1: hipLaunchKernelGGL((k_project_eig<1>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 2: hipLaunchKernelGGL((k_project_eig<2>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 3: hipLaunchKernelGGL((k_project_eig<3>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 4: hipLaunchKernelGGL((k_project_eig<4>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 5: hipLaunchKernelGGL((k_project_eig<5>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 6: hipLaunchKernelGGL((k_project_eig<6>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 7: hipLaunchKernelGGL((k_project_eig<7>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            case 8: hipLaunchKernelGGL((k_project_eig<8>), g, b, 0, stream, H, n_key, n_pool, compact, list, nl, eps, mirroring, sos, vals, c.counters.p); break;
            default: throw Error("project: unsupported block count");
        }
    }
    flush();
}
static void project_phase_c(Context& c, Context::ProjRound& R)
{
    if (R.marks.empty()) return;
    for (int part = 0; part < 2; part++)
        if (R.mark_part[part]) {
            BsrPart& m = c.part[part];
            const size_t n_pos = (size_t)std::max<int64_t>(m.ntiles * 64, m.nnzb) + 4;  // (flags are indexed like the values: by storage position; + the fill's rounding to words)
            if (m.slot_dirty.cap < n_pos) {
                m.slot_dirty.ensure(n_pos);
                MS_CHECK(hipMemsetAsync(m.slot_dirty.p, 0, m.slot_dirty.cap, c.stream));
            }
        }
    MarkBatch mb;
    mb.n = 0;
    int blocks = 0;
    auto flush_marks = [&]() {
        if (mb.n > 0) hipLaunchKernelGGL(k_proj_mark, dim3(blocks), dim3(BLOCK), 0, c.stream, mb);
        mb.n = 0;
        blocks = 0;
    };
    for (const Context::ProjRound::Mark& k : R.marks) {
        Potential& P = c.pots[(size_t)k.pot];
        BsrPart& m = c.part[P.part];
        const bool lazy = k.Hc != nullptr;
        if (k.nl <= 0) continue;
        if (mb.n == MARK_BATCH) flush_marks();
        mb.d[mb.n++] = MarkDesc{k.list, (const uint32_t*)(m.slot_of_src.p + P.kp_off), m.slot_dirty.p, k.Hc, lazy ? c.elemHf.p + P.hf_off : (float*)nullptr, k.nl, P.NB, P.n_key,
                                k.n_pool_c, P.n_pool_f, blocks, c.hf_layout};
        blocks += grid_for((int64_t)k.nl * P.NB * P.NB);
    }
    flush_marks();
    for (int part = 0; part < 2; part++)
        if (R.mark_part[part]) {
            BsrPart& m = c.part[part];
            gather_part(c, part, m.slot_dirty.p);
            fill_async(c.stream, m.slot_dirty.p, 0, ((size_t)std::max<int64_t>(m.ntiles * 64, m.nnzb) + 3) & ~(size_t)3);  // (clean for the next round)
        }
    R.marks.clear();
}
void project(Context& c, double eps, int mirroring, const uint8_t* active_host, bool by_gradient, double threshold, int* all_active, int64_t* n_projected_now,
             int64_t* n_changed_now)
{
    if (c.static_assembled) MS_CHECK(hipStreamWaitEvent(c.stream, c.aux_ev[2], 0));  // (the deltas below go into the matrix the auxiliary stream is still gathering)
    ensure_pattern(c);
    if (!c.have_hessians) throw Error("project: no element Hessians (call eval with MISTARK_EVAL_P_G_H first)");
    project_spec_discard(c);  // (a round started ahead with other parameters: its kernels first)
    project_phase_a(c, active_host, by_gradient, threshold);
    int64_t h[128];
    fetch(c, h, c.counters.p, sizeof(h));
    int64_t n_inactive = h[2], n_selected = h[3];
    if (c.world > 1) {  // the counts of the whole problem (every rank takes the same decisions)
        double mine[2] = {(double)h[2], (double)h[3]}, all[2 * 64];
        shard_allgather_scalars(c, mine, 2, all);
        n_inactive = n_selected = 0;
        for (int r = 0; r < c.world; r++) {
            n_inactive += (int64_t)all[2 * r];
            n_selected += (int64_t)all[2 * r + 1];
        }
    }
    Context::ProjRound R;
    project_phase_b(c, h, eps, mirroring, R);
    project_phase_c(c, R);
    c.n_projected_total += n_selected;
    if (n_projected_now) *n_projected_now = n_selected;
    if (n_changed_now) {  // (sharded: this rank's count, interface elements included)
        int64_t h2[2];
        fetch(c, h2, c.counters.p, sizeof(h2));
        *n_changed_now = h2[1];
    }
    if (all_active) *all_active = by_gradient ? (n_inactive == 0) : (active_host == nullptr);
}

// ---- a projection round started AHEAD of the solve that may need it ------------------------------------------------------------------------
// Progressive projection (NewtonsMethod.cpp:254-386) retries a failed solve with more elements projected, and what it will project is known
// before the solve starts: the rows whose gradient exceeds the NEXT threshold (this one times the tightening factor), at the unchanged iterate.
// On configs[3] 51 of 71 solves fail (indefinite barrier Hessians; the reference takes the same retries), and the round between two solves —
// selection, read-back, the eigen kernels' dependent chains (116 us), marks, gather — was 0.27 ms of idle solver. project_speculate runs
// phases A and B of that round on a stream of their own WHILE the solve runs (they touch pools, lists and counters, nothing the PCG reads);
// pcg()'s wait loop calls project_spec_poll, which launches phase B once phase A's counts have reached the host (a kernel writes them to pinned
// memory: no synchronisation). If the solve fails, project_spec_adopt lets the main stream wait for that work and runs phase C: the matrix the
// next solve sees is the one project() would have produced, bit for bit (same selection, same projected blocks, same ordered gather). If the
// solve succeeds, the round is dropped (its selection flags die with the next evaluation's reset).
__global__ __launch_bounds__(128) void k_spec_publish(const int64_t* __restrict__ counters, int64_t* __restrict__ dst_host, uint32_t* __restrict__ flag_host, uint32_t seq)
{
    dst_host[threadIdx.x] = counters[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_store_n(flag_host, seq, __ATOMIC_RELEASE);
        __threadfence_system();
    }
}
// (the request is left by the Newton loop BEFORE the solve and taken up by pcg() once its first batches are queued: the host's work for
// phase A — a dozen launches and a descriptor upload — then overlaps the solve's first iterations instead of delaying them)
static void project_spec_init(Context& c);
void project_speculate_request(Context& c, double eps, int mirroring, double threshold)
{
    Context::ProjSpec& S = c.spec;
    project_spec_discard(c);
    S.pending = threshold > 0.0 && project_can_speculate(c);
    S.p_eps = eps;
    S.p_mirroring = mirroring;
    S.p_threshold = threshold;
    if (S.pending) {
        project_spec_init(c);
        MS_CHECK(hipEventRecord(S.ev_in, c.stream));  // (what the round may start behind: the matrix, the pools and the gradient as they are NOW, before the solve's launches)
    }
}
void project_speculate_pending(Context& c)
{
    Context::ProjSpec& S = c.spec;
    if (!S.pending) return;
    S.pending = false;
    project_speculate(c, S.p_eps, S.p_mirroring, S.p_threshold, /*ev_in_recorded=*/true);
}
// MEASURED on configs[3] (51 of 71 solves fail and are retried) and OFF by default (option "proj_speculation"): the round's own time leaves the
// projection stage (9.9 -> 4.0 ms over 20 Newton iterations) and comes back in the solves (82.0 -> 91.0 ms: +0.12 ms per solve, whichever
// stream carries the round, also one of lowest priority) — a solve is a chain of dependent launches that each fill the chip; a kernel that runs
// beside it delays the chain by about its own duration (single PCG kernels stretched to 200-330 us under the trace), so the 0.27 ms between two
// solves are bought back at cost, and the 20 successful solves pay for rounds nobody needs: 153-155 against 156-159 Newton-steps/s. The bits are
// the same either way (tests/test_gpu_scene.py::test_projection_round_started_beside_the_solve_changes_no_bit).
bool project_can_speculate(const Context& c) { return c.world == 1 && c.proj_speculation && !c.atomic_projection && c.matrix_current && c.have_hessians && !c.dry; }
static void project_spec_init(Context& c)
{
    Context::ProjSpec& S = c.spec;
    if (!S.stream) {
        // (the stream of the early evaluation, idle while a solve runs: a FIFTH stream of the process would share a hardware queue with the main
        // stream — HIP maps streams onto four queues in creation order — and the solve's launches would queue up behind the round they are meant
        // to run beside: measured, +0.12 ms per solve)
        if (!c.pre_stream) MS_CHECK(hipStreamCreateWithFlags(&c.pre_stream, hipStreamNonBlocking));
        S.stream = c.pre_stream;
        MS_CHECK(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        MS_CHECK(hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
        MS_CHECK(hipHostMalloc((void**)&S.pinned, 130 * sizeof(int64_t), hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(S.pinned, 0, 130 * sizeof(int64_t));
    }
}
void project_speculate(Context& c, double eps, int mirroring, double threshold, bool ev_in_recorded)
{
    Context::ProjSpec& S = c.spec;
    if (!ev_in_recorded) project_spec_discard(c);
    if (!project_can_speculate(c) || !(threshold > 0.0)) return;
    project_spec_init(c);
    if (!ev_in_recorded) MS_CHECK(hipEventRecord(S.ev_in, c.stream));  // (the matrix, the pools and the gradient as the main stream leaves them)
    MS_CHECK(hipStreamWaitEvent(S.stream, S.ev_in, 0));
    hipStream_t main_stream = c.stream;
    c.stream = S.stream;
    try {
        project_phase_a(c, nullptr, true, threshold);
        S.seq++;
        hipLaunchKernelGGL(k_spec_publish, dim3(1), dim3(128), 0, c.stream, (const int64_t*)c.counters.p, S.pinned, reinterpret_cast<uint32_t*>(S.pinned + 128), S.seq);
    } catch (...) {
        c.stream = main_stream;
        throw;
    }
    c.stream = main_stream;
    S.active = true;
    S.stage = 1;
    S.threshold = threshold;
    S.eps = eps;
    S.mirroring = mirroring;
    c.n_proj_speculated++;
}
// non-blocking: phase B as soon as phase A's counts are on the host
void project_spec_poll(Context& c)
{
    Context::ProjSpec& S = c.spec;
    if (!S.active || S.stage != 1) return;
    if (__atomic_load_n(reinterpret_cast<uint32_t*>(S.pinned + 128), __ATOMIC_ACQUIRE) != S.seq) return;
    std::memcpy(S.h, S.pinned, 128 * sizeof(int64_t));
    hipStream_t main_stream = c.stream;
    c.stream = S.stream;
    try {
        project_phase_b(c, S.h, S.eps, S.mirroring, S.round);
    } catch (...) {
        c.stream = main_stream;
        throw;
    }
    c.stream = main_stream;
    MS_CHECK(hipEventRecord(S.ev_done, S.stream));
    S.stage = 2;
}
// the round the caller is about to run: taken over if it is the one started ahead (same threshold, eps and mirroring: same bits)
bool project_spec_adopt(Context& c, double eps, int mirroring, double threshold, int* all_active, int64_t* n_projected_now)
{
    Context::ProjSpec& S = c.spec;
    static const bool dbg = std::getenv("MISTARK_DEBUG_SPEC") != nullptr;
    if (dbg) std::fprintf(stderr, "[spec] adopt? active %d pending %d threshold %.17g (round: %.17g) can %d (matrix_current %d have_hessians %d)\n", (int)S.active, (int)S.pending, threshold, S.threshold,
                          (int)project_can_speculate(c), (int)c.matrix_current, (int)c.have_hessians);
    if (!S.active) return false;
    if (S.threshold != threshold || S.eps != eps || S.mirroring != mirroring || !project_can_speculate(c)) {
        project_spec_discard(c);
        return false;
    }
    while (S.stage == 1) {  // (a solve shorter than phase A: wait for the counts here)
        project_spec_poll(c);
        if (S.stage == 1) {
            __builtin_ia32_pause();
            const hipError_t q = hipStreamQuery(S.stream);
            if (q != hipErrorNotReady && q != hipSuccess) MS_CHECK(q);
        }
    }
    MS_CHECK(hipStreamWaitEvent(c.stream, S.ev_done, 0));
    project_phase_c(c, S.round);
    const int64_t n_inactive = S.h[2], n_selected = S.h[3];
    c.n_projected_total += n_selected;
    if (n_projected_now) *n_projected_now = n_selected;
    if (all_active) *all_active = n_inactive == 0;
    S.active = false;
    S.stage = 0;
    c.n_proj_adopted++;
    return true;
}
void project_spec_discard(Context& c)
{
    Context::ProjSpec& S = c.spec;
    S.pending = false;
    if (!S.active) return;
    // whatever of it is still queued or running reads the pools and the DoFs: nothing on the main stream overtakes it
    MS_CHECK(hipEventRecord(S.ev_done, S.stream));
    MS_CHECK(hipStreamWaitEvent(c.stream, S.ev_done, 0));
    S.round.marks.clear();
    S.active = false;
    S.stage = 0;
}

}  // namespace mistark
