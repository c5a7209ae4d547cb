// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the per-Newton-step hot path and their launchers.
//
//   element evaluation   : lane-per-(i,j) hyper-dual evaluation of the energy expressions (energies.hpp)
//   pattern build        : (block row, block col) keys -> radix sort -> unique -> 64-block tiles; the static part re-laid in
//                          row-aligned chunks of 8 tiles (build_aligned)
//   assembly             : element 3x3 blocks -> float BSR tiles (deterministic gather; float atomics as an option), block-Jacobi inverse
//   SpMV                 : one wavefront per chunk of complete rows, coalesced 16-B loads, in-wave segmented reduction
//   PCG                  : 3 kernels per iteration, device-resident convergence control, look-ahead batches
//   PSD projection       : per-element cyclic Jacobi eigen-decomposition; sharded runs exchange the matrix deltas


#include <chrono>

#include "kernels_common.hpp"
#include "registry.hpp"
#include "tet_closed.hpp"
#include "tri_closed.hpp"
#include "contact_closed.hpp"

namespace mistark {

// ======================================================================================================================
// Potential registry
// ======================================================================================================================
struct KindInfo
{
    const char* name;
    int NB, NBIND, NIN;
    int strides[MAX_BIND];
};
static std::vector<KindInfo> make_kinds()
{
    std::vector<KindInfo> v;
#define X(En)                                                             \
    {                                                                     \
        KindInfo k{};                                                     \
        k.name = En::name;                                                \
        k.NB = En::NB;                                                    \
        k.NBIND = En::Layout::NBIND;                                      \
        k.NIN = En::Layout::NIN;                                          \
        static_assert(En::Layout::NBIND <= MAX_BIND, "too many bindings"); \
        static_assert(En::NB <= MAX_NB, "too many DoF blocks");           \
        En::Layout::strides(k.strides);                                   \
        v.push_back(k);                                                   \
    }
    MISTARK_FOR_EACH_ENERGY(X)
#undef X
    return v;
}
static const std::vector<KindInfo>& kinds()
{
    static const std::vector<KindInfo> k = make_kinds();
    return k;
}
int n_kinds() { return (int)kinds().size(); }
const char* kind_name(int kind) { return kinds()[kind].name; }
int find_kind(const char* name)
{
    for (int i = 0; i < n_kinds(); i++)
        if (std::strcmp(kinds()[i].name, name) == 0) return i;
    return -1;
}
int kind_nb(int kind) { return kinds()[kind].NB; }
int kind_nbind(int kind) { return kinds()[kind].NBIND; }
void kind_strides(int kind, int* out) { std::memcpy(out, kinds()[kind].strides, sizeof(int) * kinds()[kind].NBIND); }

// ======================================================================================================================
// Element evaluation
// ======================================================================================================================
template <class En>
__device__ __forceinline__ void gather_inputs(const PotArgs& a, int e, double* in)
{
    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
    En::Layout::for_each([&](int b, int S, int o) {
        const int col = a.conn_col[b];
        const size_t idx = col < 0 ? 0 : (size_t)ce[col];
        const double* src = a.arr[b] + idx * S;
#pragma unroll
        for (int c = 0; c < S; c++) in[o + c] = src[c];
    });
}

// Energy only: one lane per element
template <class En>
__global__ __launch_bounds__(BLOCK) void k_eval_p(PotArgs a, double* __restrict__ elemE)
{
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    if (le >= a.e_count) return;
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    if (!element_active<En>(in)) {  // conditional potential, element switched off (SecondOrderCompiledPotential.cpp:185-197)
        elemE[pe] = 0.0;
        return;
    }
    Loader<double> L{in};
    elemE[pe] = energy_here(a, e) ? En::energy(L) : 0.0;
}

// The energies of ALL small potentials of a line-search evaluation in one launch (rigid-body terms, contact and friction tables: a dozen
// kernels of 5-7 us each, one after the other, 66 of the 125 us of an energy evaluation of configs[3]): the workgroup index picks the potential.
struct MultiP
{
    PotArgs a;
    double* E;
    int kind, pad;
};
constexpr int MULTI_P_MAX = 48;
struct MultiFirst
{
    int b[MULTI_P_MAX + 1];
    int n;
};
template <class En>
__device__ __forceinline__ void eval_p_body(const PotArgs& a, double* __restrict__ elemE, int le)
{
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    if (!element_active<En>(in)) {
        elemE[pe] = 0.0;
        return;
    }
    Loader<double> L{in};
    elemE[pe] = energy_here(a, e) ? En::energy(L) : 0.0;
}
__global__ __launch_bounds__(BLOCK) void k_eval_p_multi(const MultiP* __restrict__ descs, MultiFirst first)
{
    int d = 0;
    while (d + 1 < first.n && (int)blockIdx.x >= first.b[d + 1]) d++;
    const MultiP& D = descs[d];
    const int le = ((int)blockIdx.x - first.b[d]) * BLOCK + threadIdx.x;
    if (le >= D.a.e_count) return;
    int k = 0;
#define X(En)                              \
    if (D.kind == k) {                     \
        eval_p_body<En>(D.a, D.E, le);     \
        return;                            \
    }                                      \
    k++;
    MISTARK_FOR_EACH_ENERGY(X)
#undef X
}

// Energy + gradient + Hessian: one lane per (element, i<=j) pair of local DoFs.
// Element Hessians are stored per potential as H[a*NB+b][e][3][3]: one 72-byte row-major 3x3 block per (block pair, element);
// consecutive elements are contiguous (coalescing-friendly stores) and assembly gathers whole 72-byte blocks.
template <class En, bool STORE_H>
__device__ __forceinline__ void eval_pgh_body(const PotArgs& a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad, long long t, int hot_way)
{
    constexpr int NB = En::NB, n = 3 * NB, NP = n * (n + 1) / 2;
    if (t >= (long long)a.e_count * NP) return;
    const int le = (int)(t / NP);
    const int e = elem_of(a, le), pe = pool_of(a, le);
    int rem = (int)(t - (long long)le * NP);
    const bool first = rem == 0;
    int i = 0;
    while (rem >= n - i) {
        rem -= n - i;
        i++;
    }
    const int j = i + rem;
    if (!STORE_H && i != j) return;
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    const bool on = element_active<En>(in);
    Loader<HDual> L{in, i, j};
    const HDual r = on ? En::energy(L) : HDual(0.0);
    const int ba = i / 3, ii = i - 3 * ba, bb = j / 3, jj = j - 3 * bb;
    if (STORE_H) {
        elemH[((size_t)(ba * NB + bb) * a.n_pool + pe) * 9 + ii * 3 + jj] = r.ab;
        elemH[((size_t)(bb * NB + ba) * a.n_pool + pe) * 9 + jj * 3 + ii] = r.ab;
    }
    if (i == j && a.gpool) {  // node gradients to the pool, summed per block row in list order by k_grad_gather (an element switched off: zeros)
        a.gpool[((size_t)ba * a.n_gpool + pe) * 3 + ii] = on ? r.a : 0.0;
    } else if (i == j && on) {
        const int node = a.conn[(size_t)e * a.conn_stride + a.dof_col[ba]];
        if (a.hot_base[ba] >= 0) atomicAdd(&a.grad_hot[((size_t)hot_way * a.n_hot + a.hot_base[ba] + node) * 3 + ii], r.a);
        else atomicAdd(&grad[3 * (size_t)(a.dof_row_off[ba] + node) + ii], r.a);
    }
    if (first) elemE[pe] = energy_here(a, e) ? r.v : 0.0;
}
template <class En, bool STORE_H>
__global__ __launch_bounds__(BLOCK) void k_eval_pgh(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    eval_pgh_body<En, STORE_H>(a, elemE, elemH, grad, (long long)blockIdx.x * BLOCK + threadIdx.x, (int)(blockIdx.x & (HOT_WAYS - 1)));
}
// Energy, gradient and Hessian of ALL contact and friction tables of an evaluation in one launch (configs[3]: seven tables of a few hundred rows,
// 11-22 us each — latency of one dependent hyper-dual chain — one after the other on the auxiliary stream: 112 us, the longer leg of the
// evaluation once the tet kernel runs at two waves per SIMD). The workgroup index picks the table, as in k_eval_p_multi. Only tables: their node
// gradients go to pools (dyn_grad_gather sums them in sorted order), so nothing here adds to a row another table adds to.
struct MultiPGH
{
    PotArgs a;
    double* E;
    double* H;
    int kind, pad;
};
constexpr int FIRST_CONTACT_KIND = 0
#define X(En) +1
    MISTARK_FOR_EACH_ENERGY(X) - (0 MISTARK_FOR_EACH_CONTACT_ENERGY(X));
#undef X
__global__ __launch_bounds__(BLOCK) void k_eval_pgh_multi(const MultiPGH* __restrict__ descs, MultiFirst first, double* __restrict__ grad)
{
    int d = 0;
    while (d + 1 < first.n && (int)blockIdx.x >= first.b[d + 1]) d++;
    const MultiPGH& D = descs[d];
    const int lb = (int)blockIdx.x - first.b[d];
    const long long t = (long long)lb * BLOCK + threadIdx.x;
    int k = FIRST_CONTACT_KIND;
#define X(En)                                                                       \
    if (D.kind == k) {                                                              \
        eval_pgh_body<En, true>(D.a, D.E, D.H, grad, t, lb & (HOT_WAYS - 1));       \
        return;                                                                     \
    }                                                                               \
    k++;
    MISTARK_FOR_EACH_CONTACT_ENERGY(X)
#undef X
}
// hot rows: the HOT_WAYS partial sums in fixed order, added to what the in-place accumulating kernels (closed-form tets) left there
__global__ __launch_bounds__(BLOCK) void k_fold_hot(const double* __restrict__ grad_hot, const int32_t* __restrict__ hot_rows, int n_hot, double* __restrict__ grad)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= 3 * n_hot) return;
    double acc = 0.0;
    for (int w = 0; w < HOT_WAYS; w++) acc += grad_hot[(size_t)w * 3 * n_hot + t];
    const int r = t / 3;
    grad[3 * (size_t)hot_rows[r] + (t - 3 * r)] += acc;
}

// Contact and friction potentials in closed form (contact_closed.hpp): one lane per contact; output in k_eval_pgh's layout
struct ClosedOut
{
    const PotArgs& a;
    int e;
    size_t pe;
    double* elemH;
    double* grad;
    __device__ __forceinline__ void put_grad(int b, const V3<double>& g) const
    {
        if (a.gpool) {
            double* o = a.gpool + ((size_t)b * a.n_gpool + pe) * 3;
            o[0] = g.x; o[1] = g.y; o[2] = g.z;
            return;
        }
        const int node = a.conn[(size_t)e * a.conn_stride + a.dof_col[b]];
        double* o = a.hot_base[b] >= 0 ? &a.grad_hot[((size_t)(blockIdx.x & (HOT_WAYS - 1)) * a.n_hot + a.hot_base[b] + node) * 3] : &grad[3 * (size_t)(a.dof_row_off[b] + node)];
        atomicAdd(o, g.x);
        atomicAdd(o + 1, g.y);
        atomicAdd(o + 2, g.z);
    }
    __device__ __forceinline__ void put_block(int NB, int ba, int bb, const M3<double>& B) const
    {
        double* o = elemH + ((size_t)(ba * NB + bb) * a.n_pool + pe) * 9;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[3 * i + j] = B.m[i][j];
        if (ba != bb) {
            double* t = elemH + ((size_t)(bb * NB + ba) * a.n_pool + pe) * 9;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) t[3 * i + j] = B.m[j][i];
        }
    }
};
template <class En, bool STORE_H>
__global__ __launch_bounds__(BLOCK) void k_eval_contact_closed(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    static_assert(!HasCond<En>::value, "conditional potentials go through the generic kernel");
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    if (le >= a.e_count) return;
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    const ClosedOut out{a, e, (size_t)pe, elemH, grad};
    const double E = closed_t<En>::template eval<STORE_H>(in, out);
    elemE[pe] = energy_here(a, e) ? E : 0.0;
}

// grad[row] += sum of the pooled node gradients incident on the row, in list order (PotArgs::gpool). One lane per (block row, component),
// eight loads in flight per lane.
__global__ __launch_bounds__(BLOCK) void k_grad_gather(const double* __restrict__ gpool, const uint32_t* __restrict__ inc_start, const uint32_t* __restrict__ inc, int64_t nbr,
                                                      double* __restrict__ grad)
{
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= 3 * nbr) return;
    const int64_t row = t / 3;
    const int comp = (int)(t - 3 * row);
    const uint32_t k0 = inc_start[row], k1 = inc_start[row + 1];
    if (k0 == k1 || k1 - k0 > GRAD_LONG_ROW) return;  // (long rows: k_grad_gather_long)
    double acc = 0.0;
    for (uint32_t kb = k0; kb < k1; kb += 8) {
        uint32_t src[8];
        double h[8];
#pragma unroll
        for (int u = 0; u < 8; u++) src[u] = kb + u < k1 ? inc[kb + u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 8; u++) h[u] = gpool[src[u] != 0xFFFFFFFFu ? (size_t)src[u] * 3 + comp : (size_t)0];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += src[u] != 0xFFFFFFFFu ? h[u] : 0.0;
    }
    // (atomic: the small potentials of the same evaluation run on another stream and add to the same rows with atomics; one addition per row
    // and potential here, so rows that only closed-form elements touch keep their bits from run to run)
    atomicAdd(&grad[t], acc);
}
// Rows with more than GRAD_LONG_ROW incidences (a rigid body attached to hundreds of points): one wavefront per row, lanes stride over the
// list, fixed-order wavefront reduction: deterministic like the short rows.
__global__ __launch_bounds__(BLOCK) void k_grad_gather_long(const double* __restrict__ gpool, const uint32_t* __restrict__ inc_start, const uint32_t* __restrict__ inc,
                                                           const uint32_t* __restrict__ long_rows, int n_long, double* __restrict__ grad)
{
    const int w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_long) return;
    const uint32_t row = long_rows[w];
    const uint32_t k0 = inc_start[row], k1 = inc_start[row + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (uint32_t k = k0 + lane; k < k1; k += 64) {
        const double* g = gpool + (size_t)inc[k] * 3;
        a0 += g[0];
        a1 += g[1];
        a2 += g[2];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a0 += __shfl_down(a0, d, 64);
        a1 += __shfl_down(a1, d, 64);
        a2 += __shfl_down(a2, d, 64);
    }
    if (lane == 0) {
        atomicAdd(&grad[3 * (size_t)row], a0);
        atomicAdd(&grad[3 * (size_t)row + 1], a1);
        atomicAdd(&grad[3 * (size_t)row + 2], a2);
    }
}
// Closed-form tet kernels (tet_closed.hpp): one lane per tet; gradient through the pool above (or 12 atomics). The 16 Hessian blocks of a tet belong to 16 pools
// (H[pair][element][9]); a lane storing its own 72 bytes would make every store instruction touch 64 separate segments, so each block
// goes through LDS: the wavefront's 64 blocks of one pair are 4608 contiguous bytes and leave as nine fully coalesced stores (and nine
// more for the transposed pair).
// Energy and node gradients of a lane's tet, stored as soon as the closed form has them (before its Hessian blocks: 26 registers less to carry through
// the block loop): node gradients to the gradient pool, summed per block row by k_grad_gather, or, without a pool, 12 atomics
struct TetEarlyOut
{
    const PotArgs* a;
    double* elemE;
    double* grad;
    int e, le;
    bool valid;
    __device__ __forceinline__ void store(double E, const double* g) const
    {
        if (!valid) return;
        const PotArgs& A = *a;
        const int pe = pool_of(A, le);
        elemE[pe] = energy_here(A, e) ? E : 0.0;
        if (A.dbg & 1) return;  // measurement switch: no gradient output
        if (A.gpool) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double* gp = A.gpool + ((size_t)k * A.n_gpool + pe) * 3;
                gp[0] = g[3 * k];
                gp[1] = g[3 * k + 1];
                gp[2] = g[3 * k + 2];
            }
            return;
        }
        const int32_t* ce = A.conn + (size_t)e * A.conn_stride;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t row = (size_t)(A.dof_row_off[k] + ce[A.dof_col[k]]);
            atomicAdd(&grad[3 * row], g[3 * k]);
            atomicAdd(&grad[3 * row + 1], g[3 * k + 1]);
            atomicAdd(&grad[3 * row + 2], g[3 * k + 2]);
        }
    }
};
struct TetBlockStagedSink
{
    TetEarlyOut early;
    bool with_early;
    __device__ __forceinline__ void energy_and_gradient(double E, const double* g) const
    {
        if (with_early) early.store(E, g);
    }
    double* stage;    // [9][64] of this wavefront
    double* Hwave;    // pool position of the wavefront's first element (pair 0)
    size_t hstride;   // doubles between pair pools
    int lane, n_valid;
    __device__ __forceinline__ void put(int a, int b, const double* blk)
    {
#pragma unroll
        for (int c = 0; c < 9; c++) stage[c * 64 + lane] = blk[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double* Hab = Hwave + (size_t)(a * 4 + b) * hstride;
        double* Hba = Hwave + (size_t)(b * 4 + a) * hstride;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int idx = k * 64 + lane, el = idx / 9, c = idx - 9 * el;
            if (el < n_valid) {
                Hab[idx] = stage[c * 64 + el];
                if (a != b) Hba[idx] = stage[((c % 3) * 3 + c / 3) * 64 + el];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
// The lazy pool: what the float BSR assembly needs and nothing more. The reference casts every element block to float before it adds it
// to the matrix (BlockedSparseMatrix.h:781-814), so the block goes to memory as 9 floats, and only the 10 blocks (a <= b) of the upper
// block triangle: the gather reads (b, a) as the transpose of (a, b). 360 bytes per tet instead of 1152. Pool layout
// Hf[pair(a,b)][element][9]; the wavefront's 64 blocks of a pair are 2304 contiguous bytes = 144 float4, staged through LDS
// element-major (stride 9 floats: conflict-free) and stored as three 16-byte-per-lane instructions.
struct TetBlockFloatSink
{
    TetEarlyOut early;
    __device__ __forceinline__ void energy_and_gradient(double E, const double* g) const { early.store(E, g); }
    float* stage;     // [64 * 9] of this wavefront
    float* Hwave;     // pool position of the wavefront's first element (pair 0); 16-byte aligned (pool stride is a multiple of 64 elements)
    size_t hstride;   // floats between pair pools
    int lane, n_valid, dbg;
    __device__ __forceinline__ void put(int a, int b, const double* blk)
    {
#pragma unroll
        for (int c = 0; c < 9; c++) stage[lane * 9 + c] = (float)blk[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* dst = Hwave + (size_t)tet_pair_index(a, b) * hstride;
        if (dbg & 2) {  // measurement switch: no global stores
        } else if (dbg & 4) {
            // ELEMENT-major pool Hf[element][pair][9] (option hf_layout = 1): the ten blocks of a tet are 360 contiguous bytes, so the blocks a
            // BSR tile gathers — those of the few dozen tets around its rows' nodes — share cache lines instead of lying in ten pair pools.
            // A wavefront's stores of one pair are 36-byte pieces 360 bytes apart; its ten puts fill the same lines within the kernel.
            if (lane < n_valid) {
                float* d = Hwave + ((size_t)lane * 10 + tet_pair_index(a, b)) * 9;
#pragma unroll
                for (int c = 0; c < 9; c++) d[c] = stage[lane * 9 + c];
            }
        } else if (n_valid == 64) {
            const float4* s4 = reinterpret_cast<const float4*>(stage);
            float4* d4 = reinterpret_cast<float4*>(dst);
            d4[lane] = s4[lane];
            d4[64 + lane] = s4[64 + lane];
            if (lane < 16) d4[128 + lane] = s4[128 + lane];
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int idx = k * 64 + lane;
                if (idx < n_valid * 9) dst[idx] = stage[idx];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
// MODE 0: energy + gradient; 1: + Hessian blocks into the double pool H[pair(4a+b)][element][9] (all 16 blocks);
//      2: + Hessian blocks into the float pool (TetBlockFloatSink); 3: Hessian blocks only, into a compact double pool (the elements a
//         projection round selected: a.elem_list = that list, pools indexed by list position)
constexpr int TET_PG = 0, TET_PGH = 1, TET_PGH_F = 2, TET_H_LIST = 3;
template <class En, bool FULL, int MODE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_eval_tet_closed(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, float* __restrict__ elemHf, double* __restrict__ grad)
{
    __shared__ double stage[MODE == TET_PG ? 1 : (BLOCK / 64) * 9 * 64];
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = le < a.e_count;
    const int lev = valid ? le : a.e_count - 1;  // (lanes past the end repeat the last element and store nothing)
    const int e = elem_of(a, lev);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    double E, g[12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int le_wave = le - lane;
    const int pe_wave = pool_of(a, le_wave);
    const TetEarlyOut early{&a, elemE, grad, e, le, valid};
    if (MODE == TET_PGH || MODE == TET_H_LIST) {
        TetBlockStagedSink sink{early, MODE == TET_PGH, stage + wave * 9 * 64, elemH + (size_t)pe_wave * 9, (size_t)a.n_pool * 9, lane, min(64, a.e_count - le_wave)};
        tet_closed_eval_to<FULL>(in, E, g, sink, true);
    } else if (MODE == TET_PGH_F) {
        TetBlockFloatSink sink{early, reinterpret_cast<float*>(stage) + wave * 9 * 64, elemHf + (size_t)pe_wave * ((a.dbg & 4) ? 90 : 9), (size_t)a.n_pool * 9, lane, min(64, a.e_count - le_wave), a.dbg};
        tet_closed_eval_to<FULL>(in, E, g, sink, true);
    } else {
        tet_closed_eval<FULL>(in, E, g, nullptr, 0, false);
        early.store(E, g);
    }
}
// Membrane triangles through their invariants (tri_closed.hpp): one lane per triangle, six hyper-dual evaluations of psi(C) instead of 45 of
// the whole energy; the nine 3x3 blocks go to the double pool as 72 contiguous bytes per lane and block (neighbouring lanes: neighbouring
// elements), the node gradients to the gradient pool (k_grad_gather) or, without one, to atomics.
template <class En, bool FULL, bool STORE_H>
__global__ __launch_bounds__(BLOCK) void k_eval_tri_closed(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    if (le >= a.e_count) return;
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    double E, g[9], H[3][3][9];
    tri_closed_eval<FULL>(in, E, g, H, STORE_H);
    elemE[pe] = energy_here(a, e) ? E : 0.0;
    if (a.gpool) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double* gp = a.gpool + ((size_t)k * a.n_gpool + pe) * 3;
            gp[0] = g[3 * k];
            gp[1] = g[3 * k + 1];
            gp[2] = g[3 * k + 2];
        }
    } else {
        const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t row = (size_t)(a.dof_row_off[k] + ce[a.dof_col[k]]);
            atomicAdd(&grad[3 * row], g[3 * k]);
            atomicAdd(&grad[3 * row + 1], g[3 * k + 1]);
            atomicAdd(&grad[3 * row + 2], g[3 * k + 2]);
        }
    }
    if (STORE_H) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double* dst = elemH + ((size_t)(i * 3 + j) * a.n_pool + pe) * 9;
#pragma unroll
                for (int c = 0; c < 9; c++) dst[c] = H[i][j][c];
            }
    }
}
// ---- gradient of the potentials with device-resident tables (Context::dyn_gpool) -----------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sm /*[4]*/);  // ("Reductions and vector helpers" below)
struct DynIncDesc
{
    const int32_t* conn;
    int stride, n_elem, NB;
    uint32_t g_off;  // first contribution
    int dof_col[MAX_NB], dof_row_off[MAX_NB];
};
// contribution g -> (block row, g)
__global__ __launch_bounds__(BLOCK) void k_dyn_inc_keys(const DynIncDesc* __restrict__ D, int n_desc, int64_t total, uint32_t* __restrict__ key, uint32_t* __restrict__ val)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= total) return;
    int k = 0;
    while (k + 1 < n_desc && g >= (int64_t)D[k + 1].g_off) k++;
    const DynIncDesc& d = D[k];
    const uint32_t l = (uint32_t)g - d.g_off;
    const int b = (int)(l / (uint32_t)d.n_elem), e = (int)(l - (uint32_t)b * (uint32_t)d.n_elem);
    key[g] = (uint32_t)(d.dof_row_off[b] + d.conn[(size_t)e * d.stride + d.dof_col[b]]);
    val[g] = (uint32_t)g;
}
constexpr int DYN_LONG_ROW = 64;
// one thread per sorted position; the thread at the head of a row's run adds the run, in order, to the gradient (runs beyond DYN_LONG_ROW:
// recorded for k_dyn_grad_gather_long)
__global__ __launch_bounds__(BLOCK) void k_dyn_grad_gather(const uint32_t* __restrict__ key, const uint32_t* __restrict__ val, int64_t total, const double* __restrict__ pool,
                                                          double* __restrict__ grad, uint32_t* __restrict__ long_list, int long_cap)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= total) return;
    const uint32_t row = key[i];
    if (i > 0 && key[i - 1] == row) return;
    int64_t lo = i, hi = total;  // first position behind the run
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (key[mid] == row) lo = mid;
        else hi = mid;
    }
    const int64_t end = hi;
    if (end - i > DYN_LONG_ROW) {
        const uint32_t at = atomicAdd(&long_list[0], 1u);
        if ((int)at < long_cap) {
            long_list[1 + 2 * at] = (uint32_t)i;
            long_list[2 + 2 * at] = (uint32_t)end;
        }
        return;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int64_t j = i; j < end; j++) {
        const double* g = pool + 3 * (size_t)val[j];
        a0 += g[0];
        a1 += g[1];
        a2 += g[2];
    }
    double* gr = grad + 3 * (size_t)row;
    gr[0] += a0;
    gr[1] += a1;
    gr[2] += a2;
}
// long runs (a rigid body under tens of thousands of contacts) in two steps: workgroups sum fixed chunks of DYN_CHUNK positions counted from the
// run's start (thread t takes the positions t, t + 256, ... of the chunk; fixed tree reduction), then one thread per run adds the chunk sums in
// order: the same bits every time, and a run of 10^5 contributions is spread over the chip instead of walked by one workgroup (0.31 ms for
// configs[2]'s floor). Chunk c of the run starting at sorted position `first` owns slot first / 64 + c of `part`: runs are longer than 64 and
// DYN_CHUNK >= 128, so the slots of different runs never meet.
constexpr int DYN_CHUNK = 1024;
__global__ __launch_bounds__(BLOCK) void k_dyn_grad_gather_long(const uint32_t* __restrict__ val, const double* __restrict__ pool, double* __restrict__ part,
                                                               const uint32_t* __restrict__ long_list, int long_cap)
{
    __shared__ double sm[4];
    const int n_long = min((int)long_list[0], long_cap);
    for (int t = 0; t < n_long; t++) {
        const uint32_t first = long_list[1 + 2 * t], end = long_list[2 + 2 * t];
        const uint32_t nch = (end - first + DYN_CHUNK - 1) / DYN_CHUNK;
        for (uint32_t ch = (blockIdx.x + gridDim.x - (uint32_t)t % gridDim.x) % gridDim.x; ch < nch; ch += gridDim.x) {
            const uint32_t lo = first + ch * DYN_CHUNK, hi = min(end, lo + DYN_CHUNK);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (uint32_t j = lo + threadIdx.x; j < hi; j += BLOCK) {
                const double* g = pool + 3 * (size_t)val[j];
                a0 += g[0];
                a1 += g[1];
                a2 += g[2];
            }
            a0 = block_sum(a0, sm);
            __syncthreads();
            a1 = block_sum(a1, sm);
            __syncthreads();
            a2 = block_sum(a2, sm);
            __syncthreads();
            if (threadIdx.x == 0) {
                double* o = part + 3 * ((size_t)(first >> 6) + ch);
                o[0] = a0;
                o[1] = a1;
                o[2] = a2;
            }
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_dyn_grad_fold_long(const uint32_t* __restrict__ key, const double* __restrict__ part, double* __restrict__ grad,
                                                             const uint32_t* __restrict__ long_list, int long_cap)
{
    const int n_long = min((int)long_list[0], long_cap);
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_long) return;
    const uint32_t first = long_list[1 + 2 * t], end = long_list[2 + 2 * t];
    const uint32_t nch = (end - first + DYN_CHUNK - 1) / DYN_CHUNK;
    const double* o = part + 3 * (size_t)(first >> 6);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (uint32_t ch = 0; ch < nch; ch++) {
        a0 += o[3 * ch];
        a1 += o[3 * ch + 1];
        a2 += o[3 * ch + 2];
    }
    double* gr = grad + 3 * (size_t)key[first];
    gr[0] += a0;
    gr[1] += a1;
    gr[2] += a2;
}
constexpr int DYN_LONG_CAP = 4096;
// (re)build the sorted contribution lists when the tables changed, then add every row's sum to `grad`; on c.stream
static void dyn_grad_gather(Context& c, double* grad)
{
    if (c.dyn_total <= 0) return;
    const int64_t n = c.dyn_total;
    if (c.dyn_inc_version != c.dyn_tables_version) {
        c.dyn_key.ensure((size_t)n);
        c.dyn_key_alt.ensure((size_t)n);
        c.dyn_val.ensure((size_t)n);
        c.dyn_val_alt.ensure((size_t)n);
        hipLaunchKernelGGL(k_dyn_inc_keys, dim3(grid_for(n)), dim3(BLOCK), 0, c.stream, (const DynIncDesc*)c.dyn_desc.p, c.dyn_n_desc, n, c.dyn_key.p, c.dyn_val.p);
        int bits = 1;
        while (bits < 32 && (1ll << bits) <= c.nbr) bits++;
        size_t tmp = 0;
        hipcub::DoubleBuffer<uint32_t> dk(c.dyn_key.p, c.dyn_key_alt.p), dv(c.dyn_val.p, c.dyn_val_alt.p);
        MS_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, dk, dv, (int)n, 0, bits, c.stream));
        c.dyn_cub_tmp.ensure(tmp);
        MS_CHECK(hipcub::DeviceRadixSort::SortPairs(c.dyn_cub_tmp.p, tmp, dk, dv, (int)n, 0, bits, c.stream));  // (stable: equal rows keep the contribution order)
        c.dyn_sorted_key = dk.Current();
        c.dyn_sorted_val = dv.Current();
        c.dyn_inc_version = c.dyn_tables_version;
    }
    c.dyn_long.ensure(1 + 2 * (size_t)DYN_LONG_CAP);
    fill_async(c.stream, c.dyn_long.p, 0, sizeof(uint32_t));
    hipLaunchKernelGGL(k_dyn_grad_gather, dim3(grid_for(n)), dim3(BLOCK), 0, c.stream, c.dyn_sorted_key, c.dyn_sorted_val, n, (const double*)c.dyn_gpool.p, grad, c.dyn_long.p, DYN_LONG_CAP);
    c.dyn_long_part.ensure(3 * ((size_t)n / 64 + 2));
    hipLaunchKernelGGL(k_dyn_grad_gather_long, dim3(256), dim3(BLOCK), 0, c.stream, c.dyn_sorted_val, (const double*)c.dyn_gpool.p, c.dyn_long_part.p, (const uint32_t*)c.dyn_long.p,
                       DYN_LONG_CAP);
    hipLaunchKernelGGL(k_dyn_grad_fold_long, dim3(DYN_LONG_CAP / BLOCK), dim3(BLOCK), 0, c.stream, c.dyn_sorted_key, (const double*)c.dyn_long_part.p, grad,
                       (const uint32_t*)c.dyn_long.p, DYN_LONG_CAP);
}
static void launch_grad_gather(Context& c, Potential& P)
{
    if (!P.args.gpool || P.dyn_pool || (c.kernel_dbg & 1)) return;  // (dyn_pool: one gather for all device-resident tables, dyn_grad_gather)
    hipLaunchKernelGGL(k_grad_gather, dim3(grid_for(3 * c.nbr)), dim3(BLOCK), 0, c.stream, (const double*)P.gpool.p, (const uint32_t*)P.inc_start.p, (const uint32_t*)P.inc.p, c.nbr,
                       c.grad.p);
    if (P.n_inc_long > 0)
        hipLaunchKernelGGL(k_grad_gather_long, dim3((P.n_inc_long + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, c.stream, (const double*)P.gpool.p, (const uint32_t*)P.inc_start.p,
                           (const uint32_t*)P.inc.p, (const uint32_t*)P.inc_long.p, P.n_inc_long, c.grad.p);
}
template <class En, bool FULL>
static void launch_tri_closed(Context& c, Potential& P, int mode)
{
    if (P.args.e_count == 0) return;
    double* E = c.elemE.p + P.e_off;
    const dim3 g(grid_for(P.args.e_count)), b(BLOCK);
    if (mode == MISTARK_EVAL_P_G) hipLaunchKernelGGL((k_eval_tri_closed<En, FULL, false>), g, b, 0, c.stream, P.args, E, (double*)nullptr, c.grad.p);
    else hipLaunchKernelGGL((k_eval_tri_closed<En, FULL, true>), g, b, 0, c.stream, P.args, E, c.elemH.p + P.h_off, c.grad.p);
    launch_grad_gather(c, P);
}
// EnergyBendingFlat in closed form (EnergyDiscreteShells.cpp:64-92): E = k coef / 2 |s|^2 with s = sum_i K_i (x0_i + dt v_i) is quadratic in
// the velocities: dE/dv_i = k coef dt K_i s and d2E/dv_i dv_j = k coef dt^2 K_i K_j I3, a constant. One lane per hinge instead of the 78
// hyper-dual evaluations of the generic kernel (0.30 -> 0.04 ms for the 196 k hinges of a 256 x 256 cloth); a lane writes the 72
// contiguous bytes of each of its 16 blocks, neighbouring lanes the neighbouring 72.
template <bool STORE_H>
__global__ __launch_bounds__(BLOCK) void k_eval_bending_flat(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    using En = E_BendingFlat;
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    if (le >= a.e_count) return;
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    const double coef = in[28], k = in[29], dt = in[30];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double K = in[24 + i];
        s0 += K * (in[12 + 3 * i] + dt * in[3 * i]);
        s1 += K * (in[13 + 3 * i] + dt * in[3 * i + 1]);
        s2 += K * (in[14 + 3 * i] + dt * in[3 * i + 2]);
    }
    const double kc = k * coef;
    elemE[pe] = energy_here(a, e) ? 0.5 * kc * (s0 * s0 + s1 * s1 + s2 * s2) : 0.0;
    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double f = kc * dt * in[24 + i];
        if (a.gpool) {
            double* gp = a.gpool + ((size_t)i * a.n_gpool + pe) * 3;
            gp[0] = f * s0;
            gp[1] = f * s1;
            gp[2] = f * s2;
            continue;
        }
        const size_t row = (size_t)(a.dof_row_off[i] + ce[a.dof_col[i]]);
        atomicAdd(&grad[3 * row], f * s0);
        atomicAdd(&grad[3 * row + 1], f * s1);
        atomicAdd(&grad[3 * row + 2], f * s2);
    }
    if (STORE_H) {
        const double h = kc * dt * dt;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double d = h * in[24 + i] * in[24 + j];
                double* H = elemH + ((size_t)(i * 4 + j) * a.n_pool + pe) * 9;
                H[0] = d;   H[1] = 0.0; H[2] = 0.0;
                H[3] = 0.0; H[4] = d;   H[5] = 0.0;
                H[6] = 0.0; H[7] = 0.0; H[8] = d;
            }
    }
}
static void launch_bending_flat(Context& c, Potential& P, int mode)
{
    if (P.args.e_count == 0) return;
    double* E = c.elemE.p + P.e_off;
    if (mode == MISTARK_EVAL_P_G)
        hipLaunchKernelGGL((k_eval_bending_flat<false>), dim3(grid_for(P.args.e_count)), dim3(BLOCK), 0, c.stream, P.args, E, (double*)nullptr, c.grad.p);
    else
        hipLaunchKernelGGL((k_eval_bending_flat<true>), dim3(grid_for(P.args.e_count)), dim3(BLOCK), 0, c.stream, P.args, E, c.elemH.p + P.h_off, c.grad.p);
    launch_grad_gather(c, P);
}
// EnergyLumpedInertia, one lane per node (EnergyLumpedInertia.cpp:12-49). The energy is quadratic in v1 and its Hessian diagonal
// (d2E/dv1^2 = m (1 + damping dt) I3; nothing for a quasi-static group), so the six (i <= j) lanes of the generic kernel — each gathering the
// node's 23 inputs — compute three zeros and three times the same work. Here a lane gathers once and runs the SAME hyper-dual expression for
// (i, i), i = 0, 1, 2 (seeds at run time, one loop body: the instruction sequence of the generic kernel's diagonal lanes, hence its bits — a
// hand-derived gradient differs in the last bit, which the long contact / attachment runs amplify into other Newton counts, see
// tests/test_attach_by_distance.py); the off-diagonal entries are the zeros the generic lanes computed. configs[3]: 172 k nodes, 59 -> ~10 us
// inside an evaluation.
template <bool STORE_H>
__global__ __launch_bounds__(BLOCK) void k_eval_lumped_inertia(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    using En = E_LumpedInertia;
    const int le = blockIdx.x * BLOCK + threadIdx.x;
    if (le >= a.e_count) return;
    const int e = elem_of(a, le), pe = pool_of(a, le);
    double in[En::Layout::NIN];
    gather_inputs<En>(a, e, in);
    double g[3], h[3], E = 0.0;
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        Loader<HDual> L{in, i, i};
        const HDual r = En::energy(L);
        g[i] = r.a;
        h[i] = r.ab;
        if (i == 0) E = r.v;
    }
    elemE[pe] = energy_here(a, e) ? E : 0.0;
    if (a.gpool) {
        double* gp = a.gpool + (size_t)pe * 3;
        gp[0] = g[0]; gp[1] = g[1]; gp[2] = g[2];
    } else {
        const int node = a.conn[(size_t)e * a.conn_stride + a.dof_col[0]];
        double* o = a.hot_base[0] >= 0 ? &a.grad_hot[((size_t)(blockIdx.x & (HOT_WAYS - 1)) * a.n_hot + a.hot_base[0] + node) * 3] : &grad[3 * (size_t)(a.dof_row_off[0] + node)];
        atomicAdd(o, g[0]);
        atomicAdd(o + 1, g[1]);
        atomicAdd(o + 2, g[2]);
    }
    if (STORE_H) {
        double* H = elemH + (size_t)pe * 9;
        H[0] = h[0]; H[1] = 0.0;  H[2] = 0.0;
        H[3] = 0.0;  H[4] = h[1]; H[5] = 0.0;
        H[6] = 0.0;  H[7] = 0.0;  H[8] = h[2];
    }
}
static void launch_lumped_inertia(Context& c, Potential& P, int mode)
{
    if (P.args.e_count == 0) return;
    double* E = c.elemE.p + P.e_off;
    if (mode == MISTARK_EVAL_P_G)
        hipLaunchKernelGGL((k_eval_lumped_inertia<false>), dim3(grid_for(P.args.e_count)), dim3(BLOCK), 0, c.stream, P.args, E, (double*)nullptr, c.grad.p);
    else
        hipLaunchKernelGGL((k_eval_lumped_inertia<true>), dim3(grid_for(P.args.e_count)), dim3(BLOCK), 0, c.stream, P.args, E, c.elemH.p + P.h_off, c.grad.p);
    launch_grad_gather(c, P);
}
template <class En, bool FULL>
static void launch_tet_closed(Context& c, Potential& P, int mode, bool kernel_only = false, bool gather_only = false, double* E_override = nullptr)
{
    if (P.args.e_count == 0) return;
    double* E = E_override ? E_override : c.elemE.p + P.e_off;
    const dim3 g(grid_for(P.args.e_count)), b(BLOCK);
    struct AfterLaunch
    {
        Context& c;
        Potential& P;
        bool on;
        ~AfterLaunch()
        {
            if (on) launch_grad_gather(c, P);
        }
    } after{c, P, !kernel_only};
    if (gather_only) return;  // (the kernel ran ahead of eval(): eval_prelaunch)
    if (mode == MISTARK_EVAL_P_G) {
        hipLaunchKernelGGL((k_eval_tet_closed<En, FULL, TET_PG>), g, b, 0, c.stream, P.args, E, (double*)nullptr, (float*)nullptr, c.grad.p);
    } else if (c.lazy_active) {
        PotArgs A = P.args;
        A.n_pool = P.n_pool_f;
        if (c.hf_layout) A.dbg |= 4;  // element-major float pool
        hipLaunchKernelGGL((k_eval_tet_closed<En, FULL, TET_PGH_F>), g, b, 0, c.stream, A, E, (double*)nullptr, c.elemHf.p + P.hf_off, c.grad.p);
    } else {
        hipLaunchKernelGGL((k_eval_tet_closed<En, FULL, TET_PGH>), g, b, 0, c.stream, P.args, E, c.elemH.p + P.h_off, (float*)nullptr, c.grad.p);
    }
}
// double Hessian blocks of the listed elements of a lazy potential into a compact pool H[pair][position in list][9] of stride n_pool
void launch_tet_closed_list(Context& c, Potential& P, const uint32_t* list, int n_list, double* H, int n_pool)
{
    PotArgs A = P.args;
    A.elem_list = list;
    A.e_begin = 0;
    A.e_count = n_list;
    A.n_pool = n_pool;
    const dim3 g(grid_for(n_list)), b(BLOCK);
    if (P.name == E_TetStrain::name)
        hipLaunchKernelGGL((k_eval_tet_closed<E_TetStrain, true, TET_H_LIST>), g, b, 0, c.stream, A, (double*)nullptr, H, (float*)nullptr, (double*)nullptr);
    else
        hipLaunchKernelGGL((k_eval_tet_closed<E_TetStrainEO, false, TET_H_LIST>), g, b, 0, c.stream, A, (double*)nullptr, H, (float*)nullptr, (double*)nullptr);
}

// One lane per contact (closed form) or one lane per (contact, pair of local DoFs) (generic)? A lane of the closed form walks 1500 (deformable
// vertices only) to 4000 (rigid bodies: two jets through the quaternion update) dependent double-precision instructions: 18 to 48 us
// however short the table is, and flat up to 65 k contacts (one wavefront per SIMD). The generic kernel starts at 12 to 20 us and grows
// with the table (configs[2]: 187 us for 66 k point-triangle contacts against 48 us). Measured on configs[3], whose tables hold a few
// hundred rows each: closed forms everywhere cost 7 Newton-steps/s of 154. So the table's size decides; contact_closed_min_lanes = 0
// (tests: every table in closed form) or a lane count overrides.
template <class En>
static bool closed_contact_pays(const Context& c, int64_t n_elem)
{
    if constexpr (has_closed_contact<En>) {
        constexpr int n = 3 * En::NB, NP = n * (n + 1) / 2;
        const int64_t min_lanes = c.contact_closed_min_lanes >= 0 ? c.contact_closed_min_lanes : (En::NR > 0 ? 350000 : 100000);
        return n_elem * NP >= min_lanes;
    }
    return false;
}
template <class En>
static void launch_eval(Context& c, Potential& P, int mode)
{
    if (P.args.e_count == 0) return;
    constexpr int n = 3 * En::NB, NP = n * (n + 1) / 2;
    double* E = c.elemE.p + P.e_off;
    if (mode == MISTARK_EVAL_P) {
        PotArgs A = P.args;
        A.e_count = P.n_eown;  // (sharded: the elements whose energy counts here lead the list; P.args.e_count on one GPU)
        if (A.e_count > 0) hipLaunchKernelGGL((k_eval_p<En>), dim3(grid_for(A.e_count)), dim3(BLOCK), 0, c.stream, A, E);
    } else if (has_closed_contact<En> && !c.force_generic && !c.generic_contact && closed_contact_pays<En>(c, P.n_elem)) {  // (the whole table's size: every rank of a sharded run decides alike)
        if constexpr (has_closed_contact<En>) {
            const dim3 g(grid_for(P.args.e_count)), b(BLOCK);
            if (mode == MISTARK_EVAL_P_G) hipLaunchKernelGGL((k_eval_contact_closed<En, false>), g, b, 0, c.stream, P.args, E, (double*)nullptr, c.grad.p);
            else hipLaunchKernelGGL((k_eval_contact_closed<En, true>), g, b, 0, c.stream, P.args, E, c.elemH.p + P.h_off, c.grad.p);
        }
    } else if (mode == MISTARK_EVAL_P_G) {
        hipLaunchKernelGGL((k_eval_pgh<En, false>), dim3(grid_for((int64_t)P.args.e_count * NP)), dim3(BLOCK), 0, c.stream, P.args, E, (double*)nullptr, c.grad.p);
    } else {
        hipLaunchKernelGGL((k_eval_pgh<En, true>), dim3(grid_for((int64_t)P.args.e_count * NP)), dim3(BLOCK), 0, c.stream, P.args, E, c.elemH.p + P.h_off, c.grad.p);
    }
    if (mode != MISTARK_EVAL_P) launch_grad_gather(c, P);
}

// a contact / friction table that eval() may put into the shared launch (k_eval_pgh_multi): generic kernel, Hessian wanted; *np = lanes per row
static bool joins_multi_pgh(const Context& c, const Potential& P, int* np)
{
    if (P.kind < FIRST_CONTACT_KIND) return false;
    int k = FIRST_CONTACT_KIND;
#define X(En)                                                                                                                      \
    if (P.kind == k) {                                                                                                             \
        constexpr int n = 3 * En::NB;                                                                                              \
        *np = n * (n + 1) / 2;                                                                                                     \
        return !(has_closed_contact<En> && !c.force_generic && !c.generic_contact && closed_contact_pays<En>(c, P.n_elem));       \
    }                                                                                                                              \
    k++;
    MISTARK_FOR_EACH_CONTACT_ENERGY(X)
#undef X
    return false;
}
static void launch_eval_kind(Context& c, Potential& P, int mode)
{
    if (P.kind == KIND_CUSTOM) {  // no compiled kernel under this name: the caller supplied the expression (custom.hip)
        launch_eval_custom(c, P, mode);
        return;
    }
    if (mode != MISTARK_EVAL_P && !c.force_generic) {
        if (P.name == E_TetStrain::name) { launch_tet_closed<E_TetStrain, true>(c, P, mode); return; }
        if (P.name == E_TetStrainEO::name) { launch_tet_closed<E_TetStrainEO, false>(c, P, mode); return; }
        if (P.name == E_BendingFlat::name) { launch_bending_flat(c, P, mode); return; }
        if (P.name == E_LumpedInertia::name && !c.generic_inertia) { launch_lumped_inertia(c, P, mode); return; }
        if (P.name == E_TriangleStrain::name) { launch_tri_closed<E_TriangleStrain, true>(c, P, mode); return; }
        if (P.name == E_TriangleStrainEO::name) { launch_tri_closed<E_TriangleStrainEO, false>(c, P, mode); return; }
    }
    int k = 0;
#define X(En)                               \
    if (P.kind == k) { launch_eval<En>(c, P, mode); return; } \
    k++;
    MISTARK_FOR_EACH_ENERGY(X)
#undef X
    throw Error("unknown potential kind");
}

// ======================================================================================================================
// Reductions and vector helpers
// ======================================================================================================================
__global__ __launch_bounds__(BLOCK) void k_sum(const double* __restrict__ v, int64_t n, double* __restrict__ part)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) s += v[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(BLOCK) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ part)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) s += a[i] * b[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(BLOCK) void k_max_abs(const double* __restrict__ v, int64_t n, double* __restrict__ part)
{
    __shared__ double sm[4];
    double s = 0.0;
    // a NaN entry must not vanish in fmax (fmax(s, NaN) = s): it becomes +inf, which every later max keeps, and the Newton loop tests isfinite
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const double a = fabs(v[i]);
        s = fmax(s, a == a ? a : INFINITY);
    }
    s = block_max(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// max |v| over the listed block rows (sharded: a rank's own rows of a vector in global numbering)
__global__ __launch_bounds__(BLOCK) void k_max_abs_rows(const double* __restrict__ v, const int32_t* __restrict__ rows, int64_t n_rows, double* __restrict__ part)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < 3 * n_rows; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = t / 3;
        const double a = fabs(v[3 * (int64_t)rows[i] + (t - 3 * i)]);
        s = fmax(s, a == a ? a : INFINITY);
    }
    s = block_max(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// sum of a[row] . b[row] over the listed block rows
__global__ __launch_bounds__(BLOCK) void k_dot_rows(const double* __restrict__ a, const double* __restrict__ b, const int32_t* __restrict__ rows, int64_t n_rows, double* __restrict__ part)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < 3 * n_rows; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = t / 3, j = 3 * (int64_t)rows[i] + (t - 3 * i);
        s += a[j] * b[j];
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(BLOCK) void k_axpby(double* __restrict__ dst, double a, const double* __restrict__ x, double b, const double* __restrict__ y, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = a * x[i] + (y ? b * y[i] : 0.0);
}
__global__ __launch_bounds__(BLOCK) void k_fill(double* __restrict__ dst, double v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = v;
}

// Fills and device-to-device copies of the hot path as kernels of our own. hipMemsetAsync / hipMemcpyAsync(DeviceToDevice) go through the
// runtime's blit path (__amd_rocclr_fillBufferAligned / copyBuffer): 10-25 us of HOST time per call (profiles/r03_v5_timeline.txt: 662 fills and
// their gaps in 31 Newton iterations), which is what the launch-bound chains of a Newton iteration (contact search, contact-part pattern,
// line search) are made of. A kernel launch costs the host 4 us; several regions share one launch.
struct FillBatch
{
    uint32_t* p[FILL_BATCH_MAX];
    uint32_t n_words[FILL_BATCH_MAX];
    uint32_t value[FILL_BATCH_MAX];
    int first_block[FILL_BATCH_MAX + 1];
    int n;
};
__global__ __launch_bounds__(BLOCK) void k_fill_batch(FillBatch fb)
{
    int k = 0;
    while (k + 1 < fb.n && (int)blockIdx.x >= fb.first_block[k + 1]) k++;
    uint32_t* __restrict__ p = fb.p[k];
    const uint32_t n = fb.n_words[k], v = fb.value[k];
    const uint32_t nb = (uint32_t)(fb.first_block[k + 1] - fb.first_block[k]);
    // 16-byte stores where the region allows it
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        uint4* p4 = reinterpret_cast<uint4*>(p);
        const uint32_t n4 = n >> 2;
        const uint4 v4 = make_uint4(v, v, v, v);
        for (uint32_t i = ((uint32_t)blockIdx.x - (uint32_t)fb.first_block[k]) * BLOCK + threadIdx.x; i < n4; i += nb * BLOCK) p4[i] = v4;
        for (uint32_t i = (n4 << 2) + ((uint32_t)blockIdx.x - (uint32_t)fb.first_block[k]) * BLOCK + threadIdx.x; i < n; i += nb * BLOCK) p[i] = v;
    } else {
        for (uint32_t i = ((uint32_t)blockIdx.x - (uint32_t)fb.first_block[k]) * BLOCK + threadIdx.x; i < n; i += nb * BLOCK) p[i] = v;
    }
}
void FillQueue::add(void* p, int byte_value, size_t bytes)
{
    if (bytes == 0) return;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3) || bytes > ((size_t)1 << 33)) {  // (odd sizes: the runtime's fill)
        flush();
        MS_CHECK(hipMemsetAsync(p, byte_value, bytes, stream));
        return;
    }
    if (n == FILL_BATCH_MAX) flush();
    ptr[n] = p;
    words[n] = bytes / 4;
    const uint32_t b = (uint32_t)(byte_value & 0xff);
    value[n] = b | (b << 8) | (b << 16) | (b << 24);
    n++;
}
void FillQueue::flush()
{
    if (n == 0) return;
    FillBatch fb;
    fb.n = n;
    int blocks = 0;
    for (int k = 0; k < n; k++) {
        fb.p[k] = (uint32_t*)ptr[k];
        fb.n_words[k] = (uint32_t)words[k];
        fb.value[k] = value[k];
        fb.first_block[k] = blocks;
        blocks += (int)std::min<size_t>((words[k] / 4 + BLOCK - 1) / BLOCK + 1, 1024);
    }
    fb.first_block[n] = blocks;
    hipLaunchKernelGGL(k_fill_batch, dim3(blocks), dim3(BLOCK), 0, stream, fb);
    n = 0;
}
void fill_async(hipStream_t stream, void* p, int byte_value, size_t bytes)
{
    FillQueue q(stream);
    q.add(p, byte_value, bytes);
    q.flush();
}
__global__ __launch_bounds__(BLOCK) void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n_words)
{
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const size_t n4 = n_words >> 2;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLOCK) d4[i] = s4[i];
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += (size_t)gridDim.x * BLOCK) dst[i] = src[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += (size_t)gridDim.x * BLOCK) dst[i] = src[i];
    }
}
void copy_async(hipStream_t stream, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return;
    if ((bytes & 3) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3)) {
        MS_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
        return;
    }
    const size_t n = bytes / 4;
    const int grid = (int)std::min<size_t>((n / 4 + BLOCK - 1) / BLOCK + 1, 2048);
    hipLaunchKernelGGL(k_copy_words, dim3(grid), dim3(BLOCK), 0, stream, (const uint32_t*)src, (uint32_t*)dst, n);
}

double* host_scratch(Context& c, size_t n)
{
    if (c.h_scratch_n < n) {
        if (c.h_scratch) (void)hipHostFree(c.h_scratch);
        // (coherent + mapped: pcg() watches a control slot in here that the device writes while kernels are still running)
        MS_CHECK(hipHostMalloc((void**)&c.h_scratch, std::max<size_t>(n, 4096) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
        c.h_scratch_n = std::max<size_t>(n, 4096);
    }
    return c.h_scratch;
}
// Small device -> host read-backs (scalars, partial sums, counters: a few dozen per Newton iteration). Instead of a copy command plus a
// stream synchronisation, a one-workgroup kernel writes the words into coherent pinned host memory and then a sequence number; the
// host spins on that number. Saves the copy-engine hop and the completion-signal round trip of every read-back.
constexpr size_t PUBLISH_MAX_BYTES = 8192;
__global__ __launch_bounds__(256) void k_publish(const uint32_t* __restrict__ src, int n_words, uint32_t* __restrict__ dst_host, uint32_t* __restrict__ flag_host, uint32_t seq)
{
    for (int i = threadIdx.x; i < n_words; i += 256) dst_host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_store_n(flag_host, seq, __ATOMIC_RELEASE);
        __threadfence_system();
    }
}
// (a second pinned area for a read-back that is started, then left in flight while ANOTHER read-back runs through publish(): eval()'s partial sums
// around the contact part's pattern counts)
static bool publish_begin2(Context& c, const void* src_dev, size_t bytes)
{
    if (bytes > PUBLISH_MAX_BYTES || (bytes & 3) || (reinterpret_cast<uintptr_t>(src_dev) & 3)) return false;
    if (!c.pub2) {
        MS_CHECK(hipHostMalloc((void**)&c.pub2, PUBLISH_MAX_BYTES + 64, hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(c.pub2, 0, PUBLISH_MAX_BYTES + 64);
    }
    uint32_t* flag = reinterpret_cast<uint32_t*>(c.pub2 + PUBLISH_MAX_BYTES);
    const uint32_t seq = ++c.pub2_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, c.stream, (const uint32_t*)src_dev, (int)(bytes / 4), (uint32_t*)c.pub2, flag, seq);
    return true;
}
static void publish_end2(Context& c, void* dst_host, size_t bytes)
{
    uint32_t* flag = reinterpret_cast<uint32_t*>(c.pub2 + PUBLISH_MAX_BYTES);
    const uint32_t seq = c.pub2_seq;
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            MS_CHECK(hipStreamSynchronize(c.stream));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) throw Error("read-back kernel finished without publishing its data");
            break;
        }
    }
    std::memcpy(dst_host, c.pub2, bytes);
}
static bool publish(Context& c, void* dst_host, const void* src_dev, size_t bytes)
{
    if (bytes > PUBLISH_MAX_BYTES || (bytes & 3) || (reinterpret_cast<uintptr_t>(src_dev) & 3)) return false;
    if (!c.pub) {
        MS_CHECK(hipHostMalloc((void**)&c.pub, PUBLISH_MAX_BYTES + 64, hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(c.pub, 0, PUBLISH_MAX_BYTES + 64);
    }
    uint32_t* flag = reinterpret_cast<uint32_t*>(c.pub + PUBLISH_MAX_BYTES);
    const uint32_t seq = ++c.pub_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, c.stream, (const uint32_t*)src_dev, (int)(bytes / 4), (uint32_t*)c.pub, flag, seq);
    // spin; fall back to a real synchronisation now and then so that a failed launch surfaces as an error instead of a hang
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            MS_CHECK(hipStreamQuery(c.stream) == hipErrorNotReady ? hipSuccess : hipStreamSynchronize(c.stream));
            if (hipStreamQuery(c.stream) == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                MS_CHECK(hipStreamSynchronize(c.stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) throw Error("read-back kernel finished without publishing its data");
                break;
            }
        }
    }
    std::memcpy(dst_host, c.pub, bytes);
    return true;
}
bool host_range_pinned(Context& c, const void* host, size_t bytes)
{
    constexpr size_t MIN_BYTES = (size_t)128 << 10;  // (below: HIP's own staging path costs less than a registration is worth)
    if (!c.pin_host_arrays || c.dry || !host || bytes < MIN_BYTES) return false;
    auto it = c.pinned.find(host);
    if (it != c.pinned.end()) {
        if (it->second.bytes == bytes) return it->second.ok;
        if (it->second.ok) (void)hipHostUnregister(const_cast<void*>(host));  // (the same address at another size: registered anew)
        (void)hipGetLastError();
        c.pinned.erase(it);
    }
    const hipError_t e = hipHostRegister(const_cast<void*>(host), bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();  // (overlaps another registration, not the caller's to lock, limits: the range stays pageable)
    c.pinned[host] = Context::PinnedRange{bytes, e == hipSuccess};
    (e == hipSuccess ? c.n_pin_ok : c.n_pin_failed)++;
    return e == hipSuccess;
}
void host_range_unpin(Context& c, const void* host)
{
    auto it = c.pinned.find(host);
    if (it == c.pinned.end()) return;
    if (it->second.ok) {
        (void)hipHostUnregister(const_cast<void*>(host));  // (the caller may have freed the range already: the error is not ours to report)
        (void)hipGetLastError();
    }
    c.pinned.erase(it);
}
void h2d_staged(Context& c, void* dst_dev, const void* src_host, size_t bytes)
{
    constexpr size_t CHUNK = (size_t)4 << 20;
    if (host_range_pinned(c, src_host, bytes)) {  // page-locked in place: one direct transfer
        MS_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c.stream));
        return;
    }
    if (bytes < ((size_t)1 << 16)) {  // (small: HIP copies these through its own staging buffer without pinning anything)
        MS_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c.stream));
        return;
    }
    for (int k = 0; k < 2; k++)
        if (!c.h_stage[k]) {
            MS_CHECK(hipHostMalloc(&c.h_stage[k], CHUNK));
            MS_CHECK(hipEventCreateWithFlags(&c.h_stage_ev[k], hipEventDisableTiming));
            MS_CHECK(hipEventRecord(c.h_stage_ev[k], c.stream));
        }
    int k = 0;
    for (size_t at = 0; at < bytes; at += CHUNK, k ^= 1) {
        const size_t len = std::min(CHUNK, bytes - at);
        MS_CHECK(hipEventSynchronize(c.h_stage_ev[k]));  // the transfer that last read this area has finished
        std::memcpy(c.h_stage[k], (const char*)src_host + at, len);
        MS_CHECK(hipMemcpyAsync((char*)dst_dev + at, c.h_stage[k], len, hipMemcpyHostToDevice, c.stream));
        MS_CHECK(hipEventRecord(c.h_stage_ev[k], c.stream));
    }
}
// Small host -> device uploads of host TEMPORARIES (descriptor tables a few hundred bytes long) without waiting for the stream: the bytes are
// copied into one of four pinned slots that outlive the call, the transfer reads the slot; a slot is reused after its transfer's event (four
// uploads later: long done). The stream synchronisation this replaces drained every queued kernel at each change of the contact tables.
void h2d_small(Context& c, void* dst_dev, const void* src_host, size_t bytes)
{
    constexpr size_t SLOT = 16384;
    if (bytes == 0) return;
    if (bytes > SLOT) {
        MS_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));  // (the caller's buffer is a temporary)
        return;
    }
    const int k = c.h_small_next++ & 3;
    if (!c.h_small[k]) {
        MS_CHECK(hipHostMalloc(&c.h_small[k], SLOT));
        MS_CHECK(hipEventCreateWithFlags(&c.h_small_ev[k], hipEventDisableTiming));
    } else {
        MS_CHECK(hipEventSynchronize(c.h_small_ev[k]));
    }
    std::memcpy(c.h_small[k], src_host, bytes);
    MS_CHECK(hipMemcpyAsync(dst_dev, c.h_small[k], bytes, hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipEventRecord(c.h_small_ev[k], c.stream));
}
void fetch(Context& c, void* dst_host, const void* src_dev, size_t bytes)
{
    if (bytes == 0) return;
    if (publish(c, dst_host, src_dev, bytes)) {
        if (c.coll) c.coll->check();  // (a bounded wait of an exchange that gave up: an error, not NaNs travelling on)
        return;
    }
    if (c.h_pin_bytes < bytes) {
        if (c.h_pin) (void)hipHostFree(c.h_pin);
        c.h_pin_bytes = std::max<size_t>(bytes, 1 << 16);
        MS_CHECK(hipHostMalloc(&c.h_pin, c.h_pin_bytes));
    }
    MS_CHECK(hipMemcpyAsync(c.h_pin, src_dev, bytes, hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    std::memcpy(dst_host, c.h_pin, bytes);
    if (c.coll) c.coll->check();
}
static bool fetch_partials_begin(Context& c, int n, const double* part_dev) { return publish_begin2(c, part_dev, (size_t)n * sizeof(double)); }
static void fetch_partials_end(Context& c, int n, double* out_host, const double* part_dev, bool published)
{
    if (published) {
        publish_end2(c, out_host, (size_t)n * sizeof(double));
        return;
    }
    MS_CHECK(hipMemcpyAsync(out_host, part_dev, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
}
static void fetch_partials(Context& c, int n, double* out_host, const double* part_dev)
{
    if (publish(c, out_host, part_dev, (size_t)n * sizeof(double))) return;
    MS_CHECK(hipMemcpyAsync(out_host, part_dev, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
}
double reduce_max_abs(Context& c, const double* v, int64_t n)
{
    const int g = grid_for(n, BLOCK, VEC_GRID);
    hipLaunchKernelGGL(k_max_abs, dim3(g), dim3(BLOCK), 0, c.stream, v, n, c.partials.p);
    double* h = host_scratch(c, MAX_PARTIALS);
    fetch_partials(c, g, h, c.partials.p);
    double m = 0.0;
    for (int i = 0; i < g; i++) m = std::max(m, h[i]);
    return m;
}
double reduce_dot(Context& c, const double* a, const double* b, int64_t n)
{
    const int g = grid_for(n, BLOCK, VEC_GRID);
    hipLaunchKernelGGL(k_dot, dim3(g), dim3(BLOCK), 0, c.stream, a, b, n, c.partials.p);
    double* h = host_scratch(c, MAX_PARTIALS);
    fetch_partials(c, g, h, c.partials.p);
    double s = 0.0;
    for (int i = 0; i < g; i++) s += h[i];
    return s;
}
// two reductions, one read-back (a read-back idles the GPU for 15-60 us; the Newton loop has ~20 of them per iteration)
void reduce_dot_and_max_abs(Context& c, const double* a, const double* b, int64_t n, double* dot, double* max_abs_a)
{
    if (c.world > 1) {
        // a (the solution) is whole on every rank, b (the gradient) complete on a rank's own rows: the dot product is the sum of the ranks' shares
        const int g = grid_for(3 * c.sh.n_own, BLOCK, VEC_GRID), g2 = grid_for(n, BLOCK, VEC_GRID);
        hipLaunchKernelGGL(k_dot_rows, dim3(g), dim3(BLOCK), 0, c.stream, a, b, (const int32_t*)c.sh.grow.p, c.sh.n_own, c.partials.p);
        hipLaunchKernelGGL(k_max_abs, dim3(g2), dim3(BLOCK), 0, c.stream, a, n, c.partials.p + g);
        double* h = host_scratch(c, 2 * MAX_PARTIALS);
        fetch_partials(c, g + g2, h, c.partials.p);
        double s = 0.0, m = 0.0;
        for (int i = 0; i < g; i++) s += h[i];
        for (int i = 0; i < g2; i++) m = std::max(m, h[g + i]);
        *dot = shard_sum(c, s);
        *max_abs_a = m;
        return;
    }
    const int g = grid_for(n, BLOCK, VEC_GRID);
    hipLaunchKernelGGL(k_dot, dim3(g), dim3(BLOCK), 0, c.stream, a, b, n, c.partials.p);
    hipLaunchKernelGGL(k_max_abs, dim3(g), dim3(BLOCK), 0, c.stream, a, n, c.partials.p + g);
    double* h = host_scratch(c, 2 * MAX_PARTIALS);
    fetch_partials(c, 2 * g, h, c.partials.p);
    double s = 0.0, m = 0.0;
    for (int i = 0; i < g; i++) s += h[i];
    for (int i = 0; i < g; i++) m = std::max(m, h[g + i]);
    *dot = s;
    *max_abs_a = m;
}
static double reduce_sum(Context& c, const double* v, int64_t n)
{
    const int g = grid_for(n, BLOCK, VEC_GRID);
    hipLaunchKernelGGL(k_sum, dim3(g), dim3(BLOCK), 0, c.stream, v, n, c.partials.p);
    double* h = host_scratch(c, MAX_PARTIALS);
    fetch_partials(c, g, h, c.partials.p);
    double s = 0.0;
    for (int i = 0; i < g; i++) s += h[i];
    return s;
}
void vec_axpby(Context& c, double* dst, double a, const double* x, double b, const double* y, int64_t n)
{
    if (n == 0) return;
    // (the DoF vector: contact caches and a prelaunched evaluation are void; work vectors are nobody's input. Bound arrays change through
    // mistark_array_axpby / _fill, which say so themselves.)
    if (dst >= c.u.p && dst < c.u.p + c.ndofs) {
        c.touch();
        c.u_version++;
    }
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(n, BLOCK, 2048)), dim3(BLOCK), 0, c.stream, dst, a, x, b, y, n);
}
void vec_fill(Context& c, double* dst, double v, int64_t n)
{
    if (n == 0) return;
    if (dst >= c.u.p && dst < c.u.p + c.ndofs) {
        c.touch();
        c.u_version++;
    }
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, BLOCK, 2048)), dim3(BLOCK), 0, c.stream, dst, v, n);
}
void vec_neg(Context& c, double* dst, const double* x, int64_t n) { vec_axpby(c, dst, -1.0, x, 0.0, nullptr, n); }

// ======================================================================================================================
// prepare(): DoF layout, device arrays, kernel argument blocks, sparsity pattern
// ======================================================================================================================
// One key per element block: (block row, block column) in the numbering of the matrix this context holds (global rows on one GPU; sharded:
// local rows [0, n_own) x local columns [0, n_own + n_ghost)). Blocks whose row belongs to another rank get the key `sentinel` (= one past
// the largest real key): they sort to the end and never become a matrix block.
__global__ __launch_bounds__(BLOCK) void k_keys(PotArgs a, int NB, int n_key, uint64_t ncols, uint64_t sentinel, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t pos_off,
                                                int32_t* __restrict__ err)
{
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const int nn = NB * NB;
    if (t >= (long long)n_key * nn) return;
    const int le = (int)(t / nn);
    const int ab = (int)(t - (long long)le * nn);
    const int ba = ab / NB, bb = ab - ba * NB;
    const int e = elem_of(a, le);
    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
    int64_t ra = a.dof_row_off[ba] + ce[a.dof_col[ba]];
    int64_t rb = a.dof_row_off[bb] + ce[a.dof_col[bb]];
    uint64_t key;
    if (a.lrow) {
        ra = a.lrow[ra];
        rb = a.lrow[rb];
        if (ra < 0 || ra >= a.n_own) key = sentinel;
        else if (rb < 0) {
            key = sentinel;
            *err = 1;  // the column is neither owned nor a ghost: not registered as shared (shard_check)
        } else key = (uint64_t)ra * ncols + (uint64_t)rb;
    } else key = (uint64_t)ra * ncols + (uint64_t)rb;
    const uint32_t off = (uint32_t)ab * (uint32_t)n_key + (uint32_t)le;
    keys[pos_off + off] = key;
    idx[pos_off + off] = pos_off + off;  // key position: potential, block pair and element (make_descriptors turns it into a pool address)
}
__global__ __launch_bounds__(BLOCK) void k_diag_keys(uint64_t nrows, uint64_t ncols, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t pos_off)
{
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= nrows) return;
    keys[pos_off + r] = r * ncols + r;
    idx[pos_off + r] = NO_SRC;  // structural diagonal block, carries no data
}
__global__ __launch_bounds__(BLOCK) void k_heads(const uint64_t* __restrict__ keys, size_t n, uint64_t sentinel, uint32_t* __restrict__ head)
{
    const size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    head[k] = (keys[k] != sentinel && (k == 0 || keys[k] != keys[k - 1])) ? 1u : 0u;
}
// scan = inclusive prefix of heads. slot = scan-1.
__global__ __launch_bounds__(BLOCK) void k_slots(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ scan, size_t n,
                                                 uint64_t nbr, uint64_t sentinel, uint32_t* __restrict__ slot_of_src, uint32_t* __restrict__ colw, uint32_t* __restrict__ slot_row,
                                                 int32_t* __restrict__ diag_slot, uint32_t* __restrict__ slot_start, uint32_t* __restrict__ row_head)
{
    const size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    if (keys[k] == sentinel) {  // a block of another rank's row: no matrix block (sharded runs; sentinels sort to the end)
        if (idx[k] != NO_SRC) slot_of_src[idx[k]] = NO_SRC;
        if (k == 0 || keys[k - 1] != sentinel) slot_start[scan[k]] = (uint32_t)k;  // (scan[k] = number of blocks: closes the last block's list)
        return;
    }
    const uint32_t slot = scan[k] - 1;
    if (idx[k] != NO_SRC) slot_of_src[idx[k]] = slot;
    const bool head = (k == 0 || keys[k] != keys[k - 1]);
    if (k == n - 1) slot_start[slot + 1] = (uint32_t)n;
    if (head) {
        slot_start[slot] = (uint32_t)k;
        const uint64_t key = keys[k];
        const uint32_t row = (uint32_t)(key / nbr), col = (uint32_t)(key % nbr);
        colw[slot] = col;  // (bit 31 = last block of its row is set by k_rows)
        slot_row[slot] = row;
        row_head[slot] = (k == 0 || (uint32_t)(keys[k - 1] / nbr) != row) ? 1u : 0u;
        if (row == col) diag_slot[row] = (int32_t)slot;
    }
}
// rscan = inclusive prefix of row_head over slots: compact row of a slot = rscan-1
// (n_dev: the count lives on the device — the contact part's pattern is built without intermediate read-backs, every kernel of the chain
// is launched over the capacity and reads the actual count itself; nullptr: the host's count)
__global__ __launch_bounds__(BLOCK) void k_rows(const uint32_t* __restrict__ slot_row, const uint32_t* __restrict__ rscan, int64_t nnzb, const uint32_t* __restrict__ n_dev,
                                                int32_t* __restrict__ rowmap, int64_t* __restrict__ row_ptr, int32_t* __restrict__ tile_first_row, uint32_t* __restrict__ colw)
{
    if (n_dev) nnzb = (int64_t)*n_dev;
    const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s >= nnzb) return;
    const uint32_t r1 = rscan[s];
    const uint32_t r0 = s > 0 ? rscan[s - 1] : 0u;
    const uint32_t crow = r1 - 1;
    const bool head = r1 != r0;
    if (head) {
        rowmap[crow] = (int32_t)slot_row[s];
        row_ptr[crow] = s;
    }
    if (s == nnzb - 1) row_ptr[r1] = nnzb;
    if (s == nnzb - 1 || slot_row[s + 1] != slot_row[s]) colw[s] |= 0x80000000u;  // last block of its row
    // bit 31: the previous block (last of the previous tile) belongs to the same row
    if ((s & 63) == 0) tile_first_row[s >> 6] = (int32_t)(crow | (head ? 0u : 0x80000000u));
}

__global__ __launch_bounds__(BLOCK) void k_crow_of_row(const int32_t* __restrict__ rowmap, int64_t n_rows, const uint32_t* __restrict__ n_dev, int32_t* __restrict__ crow_of_row)
{
    if (n_dev) n_rows = (int64_t)*n_dev;
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n_rows) crow_of_row[rowmap[r]] = (int32_t)r;
}
// cnt[0 .. n_fill]: chunks per row, zero from n_rows on (n_fill = n_rows, or the capacity when the count lives on the device)
__global__ __launch_bounds__(BLOCK) void k_chunk_count(const int64_t* __restrict__ row_ptr, int64_t n_rows, const uint32_t* __restrict__ n_dev, int64_t n_fill, uint32_t* __restrict__ cnt)
{
    if (n_dev) n_rows = (int64_t)*n_dev;
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r > n_fill) return;
    const int64_t len = r < n_rows ? row_ptr[r + 1] - row_ptr[r] : 0;
    cnt[r] = len > DYN_SHORT_ROW ? (uint32_t)((len + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS) : 0u;  // short rows are summed by one lane each (spmv_chunks)
}
__global__ __launch_bounds__(BLOCK) void k_chunk_fill(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ row_chunk0, int64_t n_rows, const uint32_t* __restrict__ n_dev,
                                                      int32_t* __restrict__ chunk_row, uint32_t* __restrict__ n_chunks_out)
{
    if (n_dev) n_rows = (int64_t)*n_dev;
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r == 0 && n_chunks_out) *n_chunks_out = row_chunk0[n_rows];
    if (r >= n_rows) return;
    for (uint32_t k = row_chunk0[r]; k < row_chunk0[r + 1]; k++) chunk_row[k] = (int32_t)r;
}
__global__ void k_copy_u32(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) { *dst = *src; }
__global__ __launch_bounds__(BLOCK) void k_long_slots(const uint32_t* __restrict__ slot_start, int64_t nnzb, const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ list,
                                                     uint32_t* __restrict__ vlist, int* __restrict__ count)
{
    if (n_dev) nnzb = (int64_t)*n_dev;
    const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s >= nnzb) return;
    const uint32_t len = slot_start[s + 1] - slot_start[s];
    if (len > VERY_LONG_SLOT) vlist[atomicAdd(count + 1, 1)] = (uint32_t)s;
    else if (len > LONG_SLOT) list[atomicAdd(count, 1)] = (uint32_t)s;
}
// ---- storage of the static part: CSR order cut into row-aligned chunks -----------------------------------------------------------------
// The blocks stay in CSR order (the lanes of a tile gather neighbouring columns of the same row: few cache lines), but padding blocks
// are inserted so that every chunk of SPMV_CHUNK_TILES tiles holds complete rows only. A wavefront of the SpMV owns one chunk: nothing
// is carried in or out, no tile is read by two wavefronts, and no wavefront starts with a search for its first row (that control
// structure cost the earlier kernel 6 of 28.6 us). Rows longer than a chunk are stored after the chunks and reduced one wavefront per
// row. The layout is computed once per pattern on the host (one pass over the row lengths).
constexpr int SPMV_CHUNK_TILES = 8;  // at 2 M blocks and more; smaller matrices take shorter chunks (more wavefronts): chunk_tiles_for
__global__ __launch_bounds__(BLOCK) void k_store_fill(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ colw, const uint64_t* __restrict__ row_pos, int64_t nbr,
                                                      uint32_t* __restrict__ store_slot, uint32_t* __restrict__ scol)
{
    const int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (row >= nbr) return;
    const int64_t s0 = row_ptr[row], s1 = row_ptr[row + 1];
    const uint64_t p0 = row_pos[row];
    for (int64_t s = s0; s < s1; s++) {
        const uint32_t pos = (uint32_t)(p0 + (uint64_t)(s - s0));
        store_slot[s] = pos;
        scol[pos] = (colw[s] & 0x7fffffffu) | (s == s1 - 1 ? 0x80000000u : 0u);  // bit 31: last block of its row
    }
}
__global__ __launch_bounds__(BLOCK) void k_remap_slots(uint32_t* __restrict__ slots, size_t n, const uint32_t* __restrict__ store_slot)
{
    const size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = slots[k];
    if (s != 0xFFFFFFFFu) slots[k] = store_slot[s];
}
// a 157 k-DoF matrix in chunks of 8 tiles is 1750 wavefronts for 1024 SIMDs: 14.5 us per launch, latency-bound
// measured (us per CG iteration at 1 / 2 / 4 / 8 tiles): 0.27 M blocks 20.7 / 18.2 / 19.8 / 22.9; 0.9 M blocks 33.1 / 33.7 / 29.3 / 33.4; 2.5 M blocks: 8 tiles
static int chunk_tiles_for(int64_t nnzb) { return nnzb >= (2 << 20) ? SPMV_CHUNK_TILES : (nnzb >= (1 << 19) ? 4 : 2); }
static void build_aligned(Context& c, BsrPart& m)
{
    const int CT = c.spmv_chunk_tiles > 0 ? c.spmv_chunk_tiles : chunk_tiles_for(m.nnzb);
    m.chunk_tiles = CT;
    const int64_t nbr = c.mrows();
    std::vector<int64_t> rp((size_t)nbr + 1);
    MS_CHECK(hipMemcpyAsync(rp.data(), m.row_ptr.p, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    const uint64_t chunk = (uint64_t)CT * 64;
    std::vector<uint64_t> row_pos((size_t)nbr);
    std::vector<uint32_t> long_rows;  // rows that do not fit a chunk
    uint64_t cur = 0;
    for (int64_t r = 0; r < nbr; r++) {
        const uint64_t len = (uint64_t)(rp[r + 1] - rp[r]);
        if (len > chunk) {
            long_rows.push_back((uint32_t)r);
            cur = (cur + 63) / 64 * 64;  // the rows of a tile must be consecutive (row = first row + row ends before the lane): restart on a tile
            continue;
        }
        if (len > 0 && cur / chunk != (cur + len - 1) / chunk) cur = (cur / chunk + 1) * chunk;  // the row would straddle: pad to the next chunk
        row_pos[r] = cur;
        cur += len;
    }
    const uint64_t n_chunk_tiles = (cur + chunk - 1) / chunk * CT;
    uint64_t pos = n_chunk_tiles * 64;
    std::vector<uint64_t> long_pos;
    for (uint32_t r : long_rows) {  // long rows after the chunks, each starting on a tile
        row_pos[r] = pos;
        long_pos.push_back(pos);
        pos += ((uint64_t)(rp[r + 1] - rp[r]) + 63) / 64 * 64;
    }
    if (pos >= (1ull << 31)) throw Error("static matrix part too large");
    m.n_chunks_static = (int64_t)(n_chunk_tiles / CT);
    m.ntiles = (int64_t)(pos / 64);
    // first row of every chunk tile (bit 31: the tile starts inside a row begun in the previous tile of the same chunk)
    std::vector<int32_t> tfr((size_t)n_chunk_tiles, 0);
    {
        int64_t r = 0;
        auto is_long = [&](int64_t q) { return (uint64_t)(rp[q + 1] - rp[q]) > chunk; };
        int64_t last_row = 0;
        for (uint64_t t = 0; t < n_chunk_tiles; t++) {
            const uint64_t p = t * 64;
            // advance to the last non-long, non-empty row starting at or before p
            while (r < nbr && (is_long(r) || rp[r + 1] == rp[r] || row_pos[r] + (uint64_t)(rp[r + 1] - rp[r]) <= p)) {
                if (!is_long(r) && rp[r + 1] > rp[r]) last_row = r;
                r++;
            }
            if (r < nbr && row_pos[r] <= p) tfr[t] = (int32_t)((uint32_t)r | (row_pos[r] < p ? 0x80000000u : 0u));
            else tfr[t] = (int32_t)(uint32_t)(r < nbr ? r : last_row);  // tile starts in padding or exactly at row r
        }
    }
    m.tile_first_row.ensure(std::max<size_t>(tfr.size(), 1));
    m.row_pos.ensure((size_t)nbr);
    m.long_rows.ensure(std::max<size_t>(long_rows.size(), 1));
    m.n_long_rows = (int)long_rows.size();
    MS_CHECK(hipMemcpyAsync(m.tile_first_row.p, tfr.data(), tfr.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipMemcpyAsync(m.row_pos.p, row_pos.data(), row_pos.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
    if (!long_rows.empty()) MS_CHECK(hipMemcpyAsync(m.long_rows.p, long_rows.data(), long_rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    m.store_slot.ensure((size_t)m.nnzb);
    m.scol.ensure(std::max<size_t>((size_t)m.ntiles * 64, 1));
    m.vals.ensure(std::max<size_t>((size_t)m.ntiles * 576, 1));
    MS_CHECK(hipMemsetAsync(m.vals.p, 0, (size_t)m.ntiles * 576 * sizeof(float), c.stream));  // padding stays zero: assembly writes real blocks only
    MS_CHECK(hipMemsetAsync(m.scol.p, 0, (size_t)m.ntiles * 64 * sizeof(uint32_t), c.stream));  // padding: column 0, not a row end
    hipLaunchKernelGGL(k_store_fill, dim3(grid_for(nbr)), dim3(BLOCK), 0, c.stream, m.row_ptr.p, m.colw.p, m.row_pos.p, nbr, m.store_slot.p, m.scol.p);
    // everything that addresses vals by slot: element-block destinations and the diagonal blocks
    const size_t n_src = m.n_keys - (size_t)nbr;  // (the structural diagonal keys at the end have no source block)
    if (n_src > 0) hipLaunchKernelGGL(k_remap_slots, dim3(grid_for(n_src)), dim3(BLOCK), 0, c.stream, m.slot_of_src.p, n_src, m.store_slot.p);
    hipLaunchKernelGGL(k_remap_slots, dim3(grid_for(nbr)), dim3(BLOCK), 0, c.stream, (uint32_t*)c.diag_slot[0].p, (size_t)nbr, m.store_slot.p);
    MS_CHECK(hipStreamSynchronize(c.stream));  // (host vectors above are temporaries)
}

// Builds the sparsity pattern of one matrix part: part 0 = potentials with fixed connectivity (+ every diagonal block, so
// each block row exists), part 1 = potentials whose connectivity changes inside the Newton loop (contacts). Part 1 only
// contains the block rows it touches ("compact rows", rowmap -> global row).
static void build_pattern(Context& c, int part)
{
    BsrPart& m = c.part[part];
    c.pattern_version++;
    size_t nk = 0;
    for (auto& P : c.pots) {
        if (P.part != part) continue;
        P.kp_off = nk;
        nk += (size_t)P.n_key * P.NB * P.NB;
    }
    const size_t diag_off = nk;
    const uint64_t nrows = (uint64_t)c.mrows(), ncols = (uint64_t)c.mcols(), sentinel = nrows * ncols;
    if (part == 0) nk += (size_t)nrows;
    m.n_keys = nk;
    m.slot_of_src.ensure(std::max<size_t>(nk, 1));
    m.dirty = false;
    m.have_matrix = false;
    c.diag_slot[part].ensure((size_t)c.nbr);
    MS_CHECK(hipMemsetAsync(c.diag_slot[part].p, 0xFF, (size_t)c.nbr * sizeof(int32_t), c.stream));
    if (nk == 0) {
        m.nnzb = m.ntiles = m.n_rows = 0;
        return;
    }
    if (nk >= (1ull << 31)) throw Error("pattern too large");
    m.keys.ensure(nk);
    m.keys_alt.ensure(nk);
    m.kidx.ensure(nk + 1);
    m.kidx_alt.ensure(nk + 1);
    m.scan.ensure(nk + 1);
    for (auto& P : c.pots) {
        if (P.part != part || P.n_key == 0) continue;
        hipLaunchKernelGGL(k_keys, dim3(grid_for((int64_t)P.n_key * P.NB * P.NB)), dim3(BLOCK), 0, c.stream, P.args, P.NB, P.n_key, ncols, sentinel, m.keys.p, m.kidx.p,
                           (uint32_t)P.kp_off, c.world > 1 ? c.sh.err.p : (int32_t*)nullptr);
    }
    if (part == 0 && nrows > 0) hipLaunchKernelGGL(k_diag_keys, dim3(grid_for((int64_t)nrows)), dim3(BLOCK), 0, c.stream, nrows, ncols, m.keys.p, m.kidx.p, (uint32_t)diag_off);
    // sort (key, source) pairs
    int bits = 1;
    while (bits < 64 && (1ull << bits) <= sentinel) bits++;
    size_t tmp_bytes = 0;
    hipcub::DoubleBuffer<uint64_t> dk(m.keys.p, m.keys_alt.p);
    hipcub::DoubleBuffer<uint32_t> dv(m.kidx.p, m.kidx_alt.p);
    MS_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, dv, (int)nk, 0, bits, c.stream));
    c.cub_tmp.ensure(tmp_bytes);
    MS_CHECK(hipcub::DeviceRadixSort::SortPairs(c.cub_tmp.p, tmp_bytes, dk, dv, (int)nk, 0, bits, c.stream));
    const uint64_t* skeys = dk.Current();
    const uint32_t* sidx = dv.Current();
    uint32_t* heads = (uint32_t*)(dv.Current() == m.kidx.p ? m.kidx_alt.p : m.kidx.p);  // the other value buffer is free now
    hipLaunchKernelGGL(k_heads, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, skeys, nk, sentinel, heads);
    size_t tmp2 = 0;
    MS_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp2, heads, m.scan.p, (int)nk, c.stream));
    c.cub_tmp.ensure(tmp2);
    MS_CHECK(hipcub::DeviceScan::InclusiveSum(c.cub_tmp.p, tmp2, heads, m.scan.p, (int)nk, c.stream));
    c.counters.ensure(128);
    if (part == 1 && c.world == 1 && !c.no_bounded_pattern) {
        // ---- the contact part, rebuilt whenever the contact sets change: NO intermediate read-back. Every buffer is sized by its bound
        // (nk contributions give at most nk blocks, rows and row chunks), every kernel of the chain is launched over the bound and reads
        // the actual count on the device; the five counts reach the host in one read-back at the end (before: four round trips of
        // 25-40 us each inside a chain of tiny kernels).
        const size_t cap = nk, cap_tiles = (nk + 63) / 64;
        uint32_t* cnt = (uint32_t*)c.counters.p;  // [0] long blocks, [1] very long blocks, [2] rows, [3] row chunks, [4] blocks
        FillQueue fills(c.stream);  // (the chain's four fills in one launch, below)
        fills.add(cnt, 0, 8 * sizeof(uint32_t));
        m.colw.ensure(cap_tiles * 64);
        m.slot_row.ensure(cap);
        m.tile_first_row.ensure(cap_tiles);
        m.vals.ensure(cap_tiles * 576);
        m.slot_start.ensure(cap + 1);
        m.long_slots.ensure(cap);
        m.vlong_slots.ensure(cap / VERY_LONG_SLOT + 64);
        m.rowmap.ensure(cap);
        m.row_ptr.ensure(cap + 1);
        m.row_chunk0.ensure(cap + 2);
        m.chunk_row.ensure(2 * cap + 1);
        m.yd.ensure(3 * cap);
        m.chunk_partial.ensure(3 * (2 * cap + 1));
        m.crow_of_row.ensure((size_t)c.nbr);
        fills.add(m.colw.p, 0, cap_tiles * 64 * sizeof(uint32_t));
        uint32_t* row_head = heads;  // (heads is dead after the scan; slots beyond the last block must read 0 in the row scan)
        fills.add(row_head, 0, (nk + 1) * sizeof(uint32_t));
        fills.add(m.crow_of_row.p, 0xFF, (size_t)c.mrows() * sizeof(int32_t));
        fills.flush();
        hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(1), 0, c.stream, (const uint32_t*)(m.scan.p + (nk - 1)), cnt + 4);
        hipLaunchKernelGGL(k_slots, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, skeys, sidx, m.scan.p, nk, ncols, sentinel, m.slot_of_src.p, m.colw.p, m.slot_row.p,
                           c.diag_slot[part].p, m.slot_start.p, row_head);
        m.sorted_src = sidx;
        m.desc_lazy = -1;
        hipLaunchKernelGGL(k_long_slots, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, m.slot_start.p, (int64_t)0, (const uint32_t*)(cnt + 4), m.long_slots.p, m.vlong_slots.p, (int*)cnt);
        uint32_t* rscan = m.scan.p;  // (scan is dead after k_slots)
        size_t tmp3 = 0;
        MS_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp3, row_head, rscan, (int)nk, c.stream));
        c.cub_tmp.ensure(tmp3);
        MS_CHECK(hipcub::DeviceScan::InclusiveSum(c.cub_tmp.p, tmp3, row_head, rscan, (int)nk, c.stream));
        hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(1), 0, c.stream, (const uint32_t*)(rscan + (nk - 1)), cnt + 2);
        hipLaunchKernelGGL(k_rows, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, m.slot_row.p, rscan, (int64_t)0, (const uint32_t*)(cnt + 4), m.rowmap.p, m.row_ptr.p, m.tile_first_row.p,
                           m.colw.p);
        uint32_t* ccnt = row_head;  // reuse (nk + 1 entries)
        hipLaunchKernelGGL(k_chunk_count, dim3(grid_for(nk + 1)), dim3(BLOCK), 0, c.stream, m.row_ptr.p, (int64_t)0, (const uint32_t*)(cnt + 2), (int64_t)nk, ccnt);
        size_t tmp5 = 0;
        MS_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp5, ccnt, m.row_chunk0.p, (int)nk + 1, c.stream));
        c.cub_tmp.ensure(tmp5);
        MS_CHECK(hipcub::DeviceScan::ExclusiveSum(c.cub_tmp.p, tmp5, ccnt, m.row_chunk0.p, (int)nk + 1, c.stream));
        hipLaunchKernelGGL(k_crow_of_row, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, m.rowmap.p, (int64_t)0, (const uint32_t*)(cnt + 2), m.crow_of_row.p);
        hipLaunchKernelGGL(k_chunk_fill, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, m.row_ptr.p, m.row_chunk0.p, (int64_t)0, (const uint32_t*)(cnt + 2), m.chunk_row.p, cnt + 3);
        uint32_t h[5] = {0, 0, 0, 0, 0};
        fetch(c, h, cnt, sizeof(h));
        m.n_long = (int)h[0];
        m.n_vlong = (int)h[1];
        m.n_rows = h[2];
        m.n_chunks = h[3];
        m.nnzb = h[4];
        m.ntiles = (m.nnzb + 63) / 64;
        if (m.nnzb == 0) {
            MS_CHECK(hipMemsetAsync(m.slot_of_src.p, 0xFF, nk * sizeof(uint32_t), c.stream));
            m.n_rows = 0;
            m.n_long = m.n_vlong = 0;
            m.n_chunks = 0;
            m.n_keys = 0;
        }
        return;
    }
    uint32_t nnzb32 = 0;
    fetch(c, &nnzb32, m.scan.p + (nk - 1), sizeof(uint32_t));
    m.nnzb = nnzb32;
    m.ntiles = (m.nnzb + 63) / 64;
    if (m.nnzb == 0) {  // (sharded: nothing of this part in the rank's rows)
        MS_CHECK(hipMemsetAsync(m.slot_of_src.p, 0xFF, nk * sizeof(uint32_t), c.stream));  // no block anywhere: the projection adds no delta
        m.n_rows = 0;
        m.n_long = m.n_vlong = 0;
        m.n_chunks = 0;
        m.n_keys = 0;
        return;
    }
    m.colw.ensure((size_t)m.ntiles * 64);
    m.slot_row.ensure((size_t)m.nnzb);
    m.tile_first_row.ensure((size_t)m.ntiles);
    m.vals.ensure((size_t)m.ntiles * 576);
    m.slot_start.ensure((size_t)m.nnzb + 1);
    MS_CHECK(hipMemsetAsync(m.colw.p, 0, (size_t)m.ntiles * 64 * sizeof(uint32_t), c.stream));
    uint32_t* row_head = heads;  // (heads is dead after the scan)
    hipLaunchKernelGGL(k_slots, dim3(grid_for(nk)), dim3(BLOCK), 0, c.stream, skeys, sidx, m.scan.p, nk, ncols, sentinel, m.slot_of_src.p, m.colw.p, m.slot_row.p,
                       c.diag_slot[part].p, m.slot_start.p, row_head);
    m.sorted_src = sidx;
    m.desc_lazy = -1;  // (make_descriptors)
    // blocks with very many contributions
    m.long_slots.ensure((size_t)m.nnzb);
    c.counters.ensure(128);
    MS_CHECK(hipMemsetAsync(c.counters.p, 0, sizeof(int64_t), c.stream));
    m.vlong_slots.ensure((size_t)nk / VERY_LONG_SLOT + 64);  // (a pattern of nk contributions holds at most nk / VERY_LONG_SLOT of them)
    hipLaunchKernelGGL(k_long_slots, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, m.slot_start.p, (int64_t)m.nnzb, (const uint32_t*)nullptr, m.long_slots.p, m.vlong_slots.p, (int*)c.counters.p);
    int n_long_h[2] = {0, 0};
    // compact rows
    uint32_t* rscan = m.scan.p;  // (scan is dead after k_slots)
    size_t tmp3 = 0;
    MS_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp3, row_head, rscan, (int)m.nnzb, c.stream));
    c.cub_tmp.ensure(tmp3);
    MS_CHECK(hipcub::DeviceScan::InclusiveSum(c.cub_tmp.p, tmp3, row_head, rscan, (int)m.nnzb, c.stream));
    // the row count joins the two block counts: one read-back instead of two
    hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(1), 0, c.stream, (const uint32_t*)(rscan + (m.nnzb - 1)), (uint32_t*)c.counters.p + 2);
    int counts_h[3] = {0, 0, 0};
    fetch(c, counts_h, c.counters.p, 3 * sizeof(int));
    n_long_h[0] = counts_h[0];
    n_long_h[1] = counts_h[1];
    m.n_rows = (uint32_t)counts_h[2];
    m.n_long = n_long_h[0];
    m.n_vlong = n_long_h[1];
    m.rowmap.ensure((size_t)m.n_rows);
    m.row_ptr.ensure((size_t)m.n_rows + 1);
    hipLaunchKernelGGL(k_rows, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, m.slot_row.p, rscan, (int64_t)m.nnzb, (const uint32_t*)nullptr, m.rowmap.p, m.row_ptr.p, m.tile_first_row.p, m.colw.p);
    MS_CHECK(hipStreamSynchronize(c.stream));
    if (part == 0 && m.n_rows != c.mrows()) throw Error("internal: static part must contain every block row");
    if (part == 0) build_aligned(c, m);
    if (part == 1) {
        // row chunks of <= CHUNK_BLOCKS blocks for the chunked SpMV of the contact part (a rigid body in contact owns block rows
        // with thousands of blocks; see k_spmv_chunks)
        m.row_chunk0.ensure((size_t)m.n_rows + 1);
        uint32_t* cnt = (uint32_t*)row_head;  // reuse
        hipLaunchKernelGGL(k_chunk_count, dim3(grid_for(m.n_rows + 1)), dim3(BLOCK), 0, c.stream, m.row_ptr.p, (int64_t)m.n_rows, (const uint32_t*)nullptr, (int64_t)m.n_rows, cnt);
        size_t tmp5 = 0;
        MS_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp5, cnt, m.row_chunk0.p, (int)m.n_rows + 1, c.stream));
        c.cub_tmp.ensure(tmp5);
        MS_CHECK(hipcub::DeviceScan::ExclusiveSum(c.cub_tmp.p, tmp5, cnt, m.row_chunk0.p, (int)m.n_rows + 1, c.stream));
        uint32_t nch = 0;
        fetch(c, &nch, m.row_chunk0.p + m.n_rows, sizeof(uint32_t));
        m.n_chunks = nch;
        m.chunk_row.ensure(std::max<size_t>(nch, 1));
        m.yd.ensure(3 * std::max<size_t>((size_t)m.n_rows, 1));
        m.crow_of_row.ensure((size_t)c.nbr);
        MS_CHECK(hipMemsetAsync(m.crow_of_row.p, 0xFF, (size_t)c.mrows() * sizeof(int32_t), c.stream));
        hipLaunchKernelGGL(k_crow_of_row, dim3(grid_for(m.n_rows)), dim3(BLOCK), 0, c.stream, m.rowmap.p, (int64_t)m.n_rows, (const uint32_t*)nullptr, m.crow_of_row.p);
        m.chunk_partial.ensure(3 * std::max<size_t>(nch, 1));
        hipLaunchKernelGGL(k_chunk_fill, dim3(grid_for(m.n_rows)), dim3(BLOCK), 0, c.stream, m.row_ptr.p, m.row_chunk0.p, (int64_t)m.n_rows, (const uint32_t*)nullptr, m.chunk_row.p, (uint32_t*)nullptr);
    }
}

void prepare(Context& c)
{
    if (c.dry) throw Error("registration-only context (mistark_create_dry): nothing can be evaluated");
    if (!c.layout_dirty) return;
    c.data_version++;
    c.u_version++;
    if (c.layout_dirty) {
        // DoF layout
        int64_t off = 0;
        for (auto& s : c.dof_sets) {
            if (s.n % 3 != 0) throw Error("DoF set '" + s.label + "' size is not a multiple of 3");
            s.offset = off;
            off += s.n;
        }
        const bool resized = off != c.ndofs;
        bool queued_uploads = resized;  // anything copied from host memory in this call (decides the closing synchronisation)
        c.ndofs = off;
        c.nbr = off / 3;
        if (c.ndofs == 0) throw Error("no degrees of freedom");
        const size_t n = (size_t)c.ndofs;
        c.u.ensure(n); c.grad.ensure(n + 8); c.du.ensure(n); c.r.ensure(n); c.z.ensure(n); c.p.ensure(n); c.q.ensure(n); c.tmp_a.ensure(n); c.tmp_b.ensure(n);
        c.partials.ensure(6 * MAX_PARTIALS);
        c.ctrl.ensure(1);
        c.counters.ensure(8);
        c.active_blocks.ensure((size_t)c.nbr);
        if (resized) {
            for (auto& s : c.dof_sets)
                if (s.n > 0) MS_CHECK(hipMemcpyAsync(c.u.p + s.offset, s.host, s.n * sizeof(double), hipMemcpyHostToDevice, c.stream));
            c.part[0].dirty = c.part[1].dirty = true;
        }
        // hot rows (PotArgs::hot_base): the rows of the small DoF sets
        std::vector<int> hot_base_of_set(c.dof_sets.size(), -1);
        {
            std::vector<int32_t> hot_rows;
            for (size_t k = 0; k < c.dof_sets.size(); k++) {
                const int64_t rows = c.dof_sets[k].n / 3;
                if (rows == 0 || rows > HOT_SET_ROWS) continue;
                hot_base_of_set[k] = (int)hot_rows.size();
                for (int64_t r = 0; r < rows; r++) hot_rows.push_back((int32_t)(c.dof_sets[k].offset / 3 + r));
            }
            c.n_hot = (int)hot_rows.size();
            c.hot_rows.ensure(std::max<size_t>(hot_rows.size(), 1));
            c.grad_hot.ensure(std::max<size_t>((size_t)HOT_WAYS * 3 * hot_rows.size(), 1));
            if (!hot_rows.empty() && hot_rows != c.hot_rows_host) {  // (unchanged across the layout refreshes a change of the contact tables asks for)
                c.hot_rows_host = hot_rows;
                MS_CHECK(hipMemcpyAsync(c.hot_rows.p, c.hot_rows_host.data(), hot_rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
                queued_uploads = true;
            }
        }
        // arrays
        for (auto& a : c.arrays) {
            if (a.dof_set >= 0) {
                a.dev = c.u.p + c.dof_sets[a.dof_set].offset;
                a.need_upload = false;
            } else {
                const size_t na = (size_t)a.n_items * a.stride;
                a.own.ensure(std::max<size_t>(na, 1));
                a.dev = a.own.p;
                if (a.need_upload && na > 0 && a.host) {
                    h2d_staged(c, a.dev, a.host, na * sizeof(double));
                    a.need_upload = false;
                    queued_uploads = true;
                }
            }
        }
        // potentials, pass 1: connectivity upload and kernel argument blocks
        for (auto& P : c.pots) {
            P.lazy_capable = P.kind != KIND_CUSTOM && !c.force_generic && (P.name == E_TetStrain::name || P.name == E_TetStrainEO::name);
            // energies of node-position differences only: their Hessians annihilate the rigid translations (k_project_eig_ti)
            P.ti_projection = P.name == E_TetStrain::name || P.name == E_TetStrainEO::name || P.name == E_TriangleStrain::name || P.name == E_TriangleStrainEO::name ||
                              P.name == E_DiscreteShells::name || P.name == E_BendingFlat::name;
            if (P.conn_dirty && !P.conn_ext) {
                // every index the kernels will follow, against the size of the array it indexes (the reference would read out of bounds; here
                // the result would be a memory fault on the device): once per connectivity upload
                for (size_t b = 0; b < P.bindings.size(); b++) {
                    const mistark_binding& B = P.bindings[b];
                    if (B.conn_col < 0 || P.conn_host.empty()) continue;
                    bool seen = false;  // (several bindings usually share a column and a size)
                    for (size_t b2 = 0; b2 < b; b2++) seen = seen || (P.bindings[b2].conn_col == B.conn_col && c.arrays[P.bindings[b2].array].n_items == c.arrays[B.array].n_items);
                    if (seen) continue;
                    const int64_t n_items = c.arrays[B.array].n_items;
                    for (int64_t e = 0; e < P.n_elem; e++) {
                        const int32_t idx = P.conn_host[(size_t)e * P.conn_stride + B.conn_col];
                        if (idx < 0 || idx >= n_items)
                            throw Error("potential '" + P.name + "': connectivity entry " + std::to_string(idx) + " (element " + std::to_string(e) + ", column " + std::to_string(B.conn_col) +
                                        ") is outside the array of " + std::to_string(n_items) + " items it indexes");
                    }
                }
                P.conn.ensure(std::max<size_t>(P.conn_host.size(), 1));
                if (!P.conn_host.empty())
                    MS_CHECK(hipMemcpyAsync(P.conn.p, P.conn_host.data(), P.conn_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
                queued_uploads = true;
                P.conn_dirty = false;
                P.conn_version++;
                P.inc_sig.clear();
                c.part[P.part].dirty = true;
            }
            PotArgs& A = P.args;
            std::memset(&A, 0, sizeof(A));
            A.conn = P.conn_ext ? P.conn_ext : P.conn.p;
            A.conn_stride = P.conn_stride;
            A.n_elem = P.n_elem;
            A.n_pool = P.n_elem;
            A.elem_list = nullptr;
            A.lrow = nullptr;
            A.dbg = c.kernel_dbg;
            A.e_begin = 0;
            A.e_count = P.n_elem;
            P.n_key = P.n_elem;
            P.n_eown = P.n_elem;
            for (size_t b = 0; b < P.bindings.size(); b++) {
                const Array& arr = c.arrays[P.bindings[b].array];
                A.arr[b] = arr.dev;
                A.conn_col[b] = P.bindings[b].conn_col;
            }
            A.grad_hot = c.grad_hot.p;
            A.n_hot = c.n_hot;
            for (int k = 0; k < MAX_NB; k++) A.hot_base[k] = -1;
            // local DoF blocks: DoF sets in registration order, then binding order (SecondOrderCompiledPotential.cpp:10-33)
            int nblk = 0;
            for (int set = 0; set < (int)c.dof_sets.size(); set++)
                for (size_t b = 0; b < P.bindings.size(); b++) {
                    const Array& arr = c.arrays[P.bindings[b].array];
                    if (arr.dof_set != set) continue;
                    if (nblk >= P.NB) throw Error("potential '" + P.name + "': more DoF bindings than the kernel's " + std::to_string(P.NB) + " blocks");
                    A.dof_col[nblk] = P.bindings[b].conn_col;
                    A.dof_row_off[nblk] = (int)(c.dof_sets[set].offset / 3);
                    A.hot_base[nblk] = hot_base_of_set[set];
                    nblk++;
                }
            if (nblk != P.NB) throw Error("potential '" + P.name + "': expected " + std::to_string(P.NB) + " DoF bindings, got " + std::to_string(nblk));
        }
        // sharded runs: row partition, local numbering, the elements this rank evaluates (shard.hip)
        if (c.world > 1) shard_prepare(c);
        // one GPU: solver numbering by Morton order of the rows' positions (Context::perm_active)
        {
            const bool have_xyz = (int64_t)c.sh.coords.size() == 3 * c.nbr && c.row_order_mode == 0;
            // (the breadth-first order needs the potentials' host connectivity; a context with a handful of rows gains nothing)
            const bool want = c.world == 1 && !c.no_row_order && c.nbr >= 4096;
            int64_t conn_sig = 0;
            if (want && !have_xyz)
                for (auto& P : c.pots)
                    if (P.part == 0 && !P.conn_ext) conn_sig = conn_sig * 1000003 + (int64_t)P.conn_version * 31 + P.n_elem;
            std::vector<int64_t> sig{want ? 1 : 0, c.nbr, have_xyz ? c.sh.version : -1, conn_sig, (int64_t)c.row_order_mode};
            if (sig != c.perm_sig) {
                c.perm_sig = sig;
                if (want != c.perm_active) c.part[0].dirty = c.part[1].dirty = true;
                c.perm_active = want;
                if (want) {
                    c.iperm_h.clear();
                    if (have_xyz) {
                        const double* X = c.sh.coords.data();
                        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                        // (a row has a position only if all three coordinates are finite: NaN marks rows without one, and an inf from the caller
                        // would poison the bounding box and make the cell conversion below undefined)
                        auto positioned = [&](int64_t r) { return std::isfinite(X[3 * r]) && std::isfinite(X[3 * r + 1]) && std::isfinite(X[3 * r + 2]); };
                        for (int64_t r = 0; r < c.nbr; r++)
                            if (positioned(r))
                                for (int d = 0; d < 3; d++) {
                                    lo[d] = std::min(lo[d], X[3 * r + d]);
                                    hi[d] = std::max(hi[d], X[3 * r + d]);
                                }
                        auto spread = [](uint64_t v) {  // 21 bits -> every third bit
                            v &= 0x1fffff;
                            v = (v | v << 32) & 0x1f00000000ffffull;
                            v = (v | v << 16) & 0x1f0000ff0000ffull;
                            v = (v | v << 8) & 0x100f00f00f00f00full;
                            v = (v | v << 4) & 0x10c30c30c30c30c3ull;
                            v = (v | v << 2) & 0x1249249249249249ull;
                            return v;
                        };
                        // cells of one common edge length (the bounding box's longest edge / 1024): neighbours in space share leading bits
                        double ext = 0.0;
                        for (int d = 0; d < 3; d++) ext = std::max(ext, hi[d] - lo[d]);
                        const double inv = ext > 0.0 ? 1023.0 / ext : 0.0;
                        std::vector<std::pair<uint64_t, int32_t>> order((size_t)c.nbr);
                        for (int64_t r = 0; r < c.nbr; r++) {
                            uint64_t code = ~0ull;  // rows without a position: behind everything, in their own order
                            if (positioned(r)) {
                                code = 0;
                                for (int d = 0; d < 3; d++) code |= spread((uint64_t)std::min(1023.0, std::max(0.0, (X[3 * r + d] - lo[d]) * inv))) << d;
                            }
                            order[(size_t)r] = {code, (int32_t)r};
                        }
                        std::sort(order.begin(), order.end());
                        for (auto& o : order) c.iperm_h.push_back(o.second);
                    } else {
                        static_graph_order(c, c.iperm_h);  // no positions (the SymX shim does not know which array holds them): breadth-first
                    }
                    c.perm_h.assign((size_t)c.nbr, 0);
                    for (int64_t k = 0; k < c.nbr; k++) c.perm_h[(size_t)c.iperm_h[(size_t)k]] = (int32_t)k;
                    c.perm.ensure((size_t)c.nbr);
                    c.iperm.ensure((size_t)c.nbr);
                    MS_CHECK(hipMemcpyAsync(c.perm.p, c.perm_h.data(), (size_t)c.nbr * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
                    MS_CHECK(hipMemcpyAsync(c.iperm.p, c.iperm_h.data(), (size_t)c.nbr * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
                    queued_uploads = true;
                    c.part[0].dirty = c.part[1].dirty = true;
                }
            }
            if (c.perm_active)
                for (auto& P : c.pots) {
                    P.args.lrow = c.perm.p;
                    P.args.n_own = (int)c.nbr;
                }
        }
        // pass 2: pools (static potentials first, so their offsets do not move when only the contact tables change size); a potential's
        // pools hold n_key elements: all of them, or the rank's list
        size_t e_off = 0, h_off = 0, hf_off = 0;
        std::vector<DynIncDesc> dyn_desc;
        std::vector<std::pair<Potential*, int64_t>> dyn_goff;
        int64_t dyn_total = 0;
        for (int part = 0; part < 2; part++) {
        for (auto& P : c.pots) {
            if (P.part != part) continue;
            PotArgs& A = P.args;
            if (P.h_off != h_off || P.hf_off != hf_off) c.part[part].desc_lazy = -1;  // pool addresses moved: make_descriptors again
            P.e_off = e_off;
            P.h_off = h_off;
            P.k_off = h_off / 9;
            e_off += (size_t)P.n_key;
            P.n_pool_f = (P.n_key + 63) / 64 * 64;
            // (a lazy potential's share of the double pool is unused while the lazy path is on — except as the compact pool of its projection
            // rounds, project_phase_b: whole wavefronts of 64 elements, hence the rounding)
            h_off += (size_t)(P.lazy_capable ? P.n_pool_f : P.n_key) * 9 * P.NB * P.NB;
            P.hf_off = hf_off;
            if (P.lazy_capable) hf_off += (size_t)P.n_pool_f * 9 * 10;
            // gradient pool + incidence lists (Potential::grad_gather)
            const bool closed_tri = P.kind != KIND_CUSTOM && !c.force_generic && (P.name == E_TriangleStrain::name || P.name == E_TriangleStrainEO::name);
            // ... and the generic kernels' potentials with several nodes per element (a single node per element: one addition per row, nothing to
            // order); the rows of rigid bodies attached to many points are summed by k_grad_gather_long
            const bool generic_pool = P.kind != KIND_CUSTOM && P.NB >= 2;
            // Tables the CALLER refills inside the Newton loop (a drop-in's contact and friction potentials: part 1, host connectivity): the host-built
            // incidence lists below cost a pass over all block rows and an upload per potential and table change — 1.2 ms per evaluation at 172 k rows,
            // what made the drop-in's energy evaluations 30 times the mirror's. Their node gradients go through the pool of the device-resident
            // tables instead (sorted by block row on the device, dyn_grad_gather): same fixed order of summation, no pass over the rows.
            const bool host_dynamic = P.part == 1 && P.NB >= 2 && !P.conn_ext && !P.conn_host.empty() && P.kind != KIND_CUSTOM && !c.no_dyn_pool && !c.no_grad_gather && c.world == 1;
            P.grad_gather = (P.lazy_capable || closed_tri || generic_pool) && !P.conn_ext && !host_dynamic && !P.conn_host.empty() && !c.no_grad_gather;
            std::vector<int64_t> sig{(int64_t)P.n_elem, c.nbr, (int64_t)P.n_key, (int64_t)P.conn_version, (int64_t)(c.world > 1 ? c.sh.version_lists : 0)};
            for (int k = 0; k < P.NB; k++) {
                sig.push_back(A.dof_col[k]);
                sig.push_back(A.dof_row_off[k]);
            }
            if (!P.grad_gather) {
                A.gpool = nullptr;
                P.inc_sig.clear();
            }
            P.dyn_pool = (P.conn_ext != nullptr && P.kind != KIND_CUSTOM && !c.no_dyn_pool && !c.no_grad_gather) || host_dynamic;
            if (P.dyn_pool && P.n_key > 0) {
                DynIncDesc d{};
                d.conn = P.conn_ext ? P.conn_ext : A.conn;
                d.stride = P.conn_stride;
                d.n_elem = P.n_key;
                d.NB = P.NB;
                d.g_off = (uint32_t)dyn_total;
                for (int k = 0; k < P.NB; k++) {
                    d.dof_col[k] = A.dof_col[k];
                    d.dof_row_off[k] = A.dof_row_off[k];
                }
                dyn_desc.push_back(d);
                dyn_goff.push_back({&P, dyn_total});
                dyn_total += (int64_t)P.NB * P.n_key;
            }
            if (P.grad_gather && P.inc_sig == sig) {  // lists still valid (prepare() runs at every change of the contact sets)
                A.gpool = P.gpool.p;
                A.n_gpool = P.n_pool_f;
            } else if (P.grad_gather) {
                P.inc_sig = sig;
                const int n_gpool = P.n_pool_f;
                std::vector<uint32_t> list;  // (sharded: the rank's element list)
                if (A.elem_list) {
                    list.resize((size_t)P.n_key);
                    MS_CHECK(hipMemcpyAsync(list.data(), P.elem_list.p, list.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
                    MS_CHECK(hipStreamSynchronize(c.stream));
                }
                auto elem = [&](int le) { return A.elem_list ? (int)list[(size_t)le] : le; };
                std::vector<uint32_t> start((size_t)c.nbr + 1, 0u), inc((size_t)P.n_key * P.NB);
                for (int le = 0; le < P.n_key; le++)
                    for (int k = 0; k < P.NB; k++) start[(size_t)(A.dof_row_off[k] + P.conn_host[(size_t)elem(le) * P.conn_stride + A.dof_col[k]]) + 1]++;
                for (int64_t r = 0; r < c.nbr; r++) start[(size_t)r + 1] += start[(size_t)r];
                std::vector<uint32_t> fill(start.begin(), start.end() - 1);
                for (int le = 0; le < P.n_key; le++)  // element-major: the contributions of a row are summed in element order
                    for (int k = 0; k < P.NB; k++)
                        inc[fill[(size_t)(A.dof_row_off[k] + P.conn_host[(size_t)elem(le) * P.conn_stride + A.dof_col[k]])]++] = (uint32_t)k * (uint32_t)n_gpool + (uint32_t)le;
                std::vector<uint32_t> long_rows;
                for (int64_t r = 0; r < c.nbr; r++)
                    if (start[(size_t)r + 1] - start[(size_t)r] > (uint32_t)GRAD_LONG_ROW) long_rows.push_back((uint32_t)r);
                P.n_inc_long = (int)long_rows.size();
                P.inc_long.ensure(std::max<size_t>(long_rows.size(), 1));
                if (!long_rows.empty()) MS_CHECK(hipMemcpyAsync(P.inc_long.p, long_rows.data(), long_rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
                P.inc_start.ensure(start.size());
                P.inc.ensure(std::max<size_t>(inc.size(), 1));
                P.gpool.ensure(std::max<size_t>((size_t)n_gpool * P.NB * 3, 1));
                MS_CHECK(hipMemcpyAsync(P.inc_start.p, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
                if (!inc.empty()) MS_CHECK(hipMemcpyAsync(P.inc.p, inc.data(), inc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
                MS_CHECK(hipStreamSynchronize(c.stream));  // (host vectors are temporaries)
                A.gpool = P.gpool.p;
                A.n_gpool = n_gpool;
            }
        }
        }
        // pool of the device-resident tables' node gradients (dyn_grad_gather)
        if (dyn_total >= (1ll << 31)) throw Error("too many contact contributions");
        c.dyn_total = dyn_total;
        c.dyn_n_desc = (int)dyn_desc.size();
        c.dyn_tables_version++;
        c.dyn_gpool.ensure(std::max<size_t>(3 * (size_t)dyn_total, 1));
        for (auto& pg : dyn_goff) {
            pg.first->args.gpool = c.dyn_gpool.p + 3 * (size_t)pg.second;
            pg.first->args.n_gpool = pg.first->n_key;
        }
        if (!dyn_desc.empty()) {
            c.dyn_desc.ensure(dyn_desc.size() * sizeof(DynIncDesc));
            h2d_small(c, c.dyn_desc.p, dyn_desc.data(), dyn_desc.size() * sizeof(DynIncDesc));  // (the host vector is a temporary: through a pinned slot, no stream synchronisation)
        }
        c.n_elem_total = e_off;
        c.hess_total = h_off;
        c.hf_total = hf_off;
        c.elemE.ensure(std::max<size_t>(e_off, 1));
        // (the element-Hessian pools are allocated by the first evaluation that writes them)
        c.is_projected.ensure(std::max<size_t>(e_off, 1) + 4);  // (+4: the zero fill rounds up to whole words)
        if (c.world > 1) {  // other ranks' elements count 0
            // (kernels started ahead of the evaluation — eval_prelaunch — may have written their energies already: the fill waits for them and
            // leaves the ranges alone that still are where those kernels wrote; a kernel whose range has moved is launched again by eval())
            std::vector<std::pair<size_t, size_t>> keep;
            if (c.pre[0].valid) {  // (slot 1 writes its energies elsewhere)
                MS_CHECK(hipStreamWaitEvent(c.stream, c.pre[0].ev_out, 0));
                for (const Context::EvalPre::Item& it : c.pre[0].items) {
                    const Potential& P = c.pots[(size_t)it.pot];
                    if (it.E == (const void*)(c.elemE.p + P.e_off) && (size_t)it.args.e_count <= (size_t)P.n_key) keep.push_back({P.e_off, (size_t)it.args.e_count});
                }
                std::sort(keep.begin(), keep.end());
            }
            size_t at = 0;
            const size_t total = std::max<size_t>(e_off, 1);
            for (const auto& k : keep) {
                if (k.first > at) fill_async(c.stream, c.elemE.p + at, 0, (k.first - at) * sizeof(double));
                at = std::max(at, k.first + k.second);
            }
            if (total > at) fill_async(c.stream, c.elemE.p + at, 0, (total - at) * sizeof(double));
        }
        c.dinv.ensure((size_t)c.nbr * 9);
        // (a refresh that only followed new contact-table sizes copied nothing from the host: no reason to wait for the stream)
        if (queued_uploads) MS_CHECK(hipStreamSynchronize(c.stream));
        c.layout_dirty = false;
        c.have_hessians = false;
    }
}
// Sparsity patterns are built on first use (assembly / projection into an assembled matrix): evaluations that only need
// energies (line search) never pay for a contact-set change.
void ensure_pattern(Context& c)
{
    prepare(c);
    for (int part = 0; part < 2; part++)
        if (c.part[part].dirty) build_pattern(c, part);
}

// ======================================================================================================================
// eval()
// ======================================================================================================================
static void build_pattern(Context& c, int part);
static void build_pattern_part(Context& c, int part) { build_pattern(c, part); }
static const bool g_eval_events = std::getenv("MISTARK_EVAL_EVENTS") != nullptr;
static void evt_mark(Context& c, int k, hipStream_t s)
{
    if (!g_eval_events) return;
    if (!c.evt[k]) MS_CHECK(hipEventCreate(&c.evt[k]));
    MS_CHECK(hipEventRecord(c.evt[k], s));
    c.evt_armed[k] = true;
}
static void evt_collect(Context& c)  // (the previous evaluation's stamps: all of them are long past)
{
    if (!g_eval_events || !c.evt_armed[0]) return;
    bool any = false;
    for (int k = 1; k < 6; k++) {
        if (!c.evt_armed[k]) continue;
        float ms = 0.f;
        if (hipEventSynchronize(c.evt[k]) == hipSuccess && hipEventElapsedTime(&ms, c.evt[0], c.evt[k]) == hipSuccess) {
            c.evt_sum[k] += 1e3 * ms;
            any = true;
        }
    }
    if (any) c.evt_n++;
    for (int k = 0; k < 6; k++) c.evt_armed[k] = false;
}
void eval_prelaunch(Context& c, int mode, bool lazy)
{
    Context::EvalPre& pre = c.pre[mode == MISTARK_EVAL_P ? 0 : 1];
    if (c.no_eval_prelaunch || c.no_eval_overlap || c.layout_dirty || c.force_generic || c.kernel_dbg || c.dry || pre.valid) return;
    const bool lazy_active = mode == MISTARK_EVAL_P_G_H && lazy && !c.atomic_assembly && c.hf_total > 0;
    if (mode == MISTARK_EVAL_P_G_H && (c.elemH.cap < std::max<size_t>(c.hess_total, 1) || c.elemHf.cap < std::max<size_t>(lazy_active ? c.hf_total : 0, 16))) return;  // (first evaluation: eval() allocates)
    if (c.elemE.cap < std::max<size_t>(c.n_elem_total, 1)) return;
    const bool apart = mode != MISTARK_EVAL_P;
    if (apart && c.elemE_pre.cap < std::max<size_t>(c.n_elem_total, 1)) {
        if (c.pre[0].valid) return;  // (no allocation while kernels are in flight)
        c.elemE_pre.ensure(std::max<size_t>(c.n_elem_total, 1));
    }
    pre.items.clear();
    for (size_t pi = 0; pi < c.pots.size(); pi++) {
        Potential& P = c.pots[pi];
        if (P.kind == KIND_CUSTOM || P.args.e_count == 0 || !P.args.gpool) continue;
        if (P.name != E_TetStrain::name && P.name != E_TetStrainEO::name) continue;
        if (mode == MISTARK_EVAL_P_G_H && lazy_active != (P.lazy_capable && lazy_active)) continue;  // (a tet potential outside the lazy pool: not here)
        pre.items.push_back(Context::EvalPre::Item{(int)pi, P.args, (const void*)((apart ? c.elemE_pre.p : c.elemE.p) + P.e_off),
                                               mode != MISTARK_EVAL_P_G_H ? nullptr : (lazy_active ? (const void*)(c.elemHf.p + P.hf_off) : (const void*)(c.elemH.p + P.h_off))});
    }
    if (pre.items.empty()) return;
    if (!c.pre_stream) MS_CHECK(hipStreamCreateWithFlags(&c.pre_stream, hipStreamNonBlocking));
    if (!pre.ev_in) {
        MS_CHECK(hipEventCreateWithFlags(&pre.ev_in, hipEventDisableTiming));
        MS_CHECK(hipEventCreateWithFlags(&pre.ev_out, hipEventDisableTiming));
    }
    MS_CHECK(hipEventRecord(pre.ev_in, c.stream));  // (the DoFs of this evaluation are final on the main stream)
    MS_CHECK(hipStreamWaitEvent(c.pre_stream, pre.ev_in, 0));
    if (mode == MISTARK_EVAL_P_G_H) {
        evt_collect(c);
        evt_mark(c, 0, c.pre_stream);
    }
    hipStream_t main_stream = c.stream;
    const bool lazy_before = c.lazy_active;
    c.stream = c.pre_stream;
    c.lazy_active = lazy_active;
    try {
        for (const Context::EvalPre::Item& it : pre.items) {
            Potential& P = c.pots[(size_t)it.pot];
            if (mode == MISTARK_EVAL_P) launch_eval_kind(c, P, mode);  // (energy only: one kernel, nothing to gather)
            else if (P.name == E_TetStrain::name) launch_tet_closed<E_TetStrain, true>(c, P, mode, true, false, c.elemE_pre.p + P.e_off);
            else launch_tet_closed<E_TetStrainEO, false>(c, P, mode, true, false, c.elemE_pre.p + P.e_off);
        }
    } catch (...) {
        c.stream = main_stream;
        c.lazy_active = lazy_before;
        throw;
    }
    c.stream = main_stream;
    c.lazy_active = lazy_before;
    MS_CHECK(hipEventRecord(pre.ev_out, c.pre_stream));
    if (mode == MISTARK_EVAL_P_G_H) evt_mark(c, 1, c.pre_stream);
    pre.valid = true;
    pre.mode = mode;
    pre.lazy_active = lazy_active;
}
static double host_now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void eval(Context& c, int mode, double* E, double* grad_host, double* grad_max_abs, bool lazy)
{
    const double t_enter = host_now_s();  // (counters eval_pgh_issue_us / eval_pgh_wait_us: where an evaluation's wall time goes on the host)
    prepare(c);
    if (mode == MISTARK_EVAL_P_G_H) {
        // lazy: float upper-triangle blocks for the potentials that can recompute their double blocks on demand (Potential::lazy_capable)
        c.lazy_active = lazy && !c.atomic_assembly && c.hf_total > 0;
        c.elemH.ensure(std::max<size_t>(c.hess_total, 1));  // (the lazy potentials' share stays untouched address space)
        c.elemHf.ensure(std::max<size_t>(c.lazy_active ? c.hf_total : 0, 16));  // (the gather reads element 0 of the pool that does not apply)
    }
    // The contact part's sparsity pattern (about fifty small launches and three read-backs, all latency) is rebuilt whenever the contact sets
    // changed; its inputs are final before the evaluation starts, so it runs on a side stream while this stream evaluates the elements
    const bool overlap_pattern = mode == MISTARK_EVAL_P_G_H && c.world == 1 && c.part[1].dirty && !c.part[0].dirty && !c.no_pattern_overlap;
    if (overlap_pattern) {
        if (!c.side_stream) {
            MS_CHECK(hipStreamCreateWithFlags(&c.side_stream, hipStreamNonBlocking));
            MS_CHECK(hipEventCreateWithFlags(&c.side_ev[0], hipEventDisableTiming));
            MS_CHECK(hipEventCreateWithFlags(&c.side_ev[1], hipEventDisableTiming));
        }
        MS_CHECK(hipEventRecord(c.side_ev[0], c.stream));  // (the contact tables were written on this stream)
    }
    if (mode != MISTARK_EVAL_P) {
        FillQueue fills(c.stream);
        fills.add(c.grad.p, 0, (size_t)c.ndofs * sizeof(double));
        if (c.n_hot > 0) fills.add(c.grad_hot.p, 0, (size_t)HOT_WAYS * 3 * c.n_hot * sizeof(double));
    }
    // kernels launched ahead of this call (eval_prelaunch): whatever becomes of their results, nothing on this stream overtakes them
    // (the wait sits in front of the first launch that touches their pools, below: the small potentials of this evaluation need not wait)
    bool pre_ok = false, pre_pending = false;
    Context::EvalPre& pre = c.pre[mode == MISTARK_EVAL_P ? 0 : 1];
    if (pre.valid) {
        pre_pending = true;
        pre_ok = pre.mode == mode && (mode != MISTARK_EVAL_P_G_H || pre.lazy_active == c.lazy_active);
        if (!pre_ok) c.n_prelaunch_dropped++;
        pre.valid = false;
    }
    // The handful of large potentials (a million tets: 230 us) and the dozens of small ones (rigid bodies, the 35 contact and friction
    // tables: 5-12 us each, latency, one after the other) share nothing but the gradient, which both sides add to atomically: the small
    // ones go to their own stream and disappear behind the large ones.
    // (energy-only evaluations are too short for it: the two stream joins cost more than they hide)
    const bool split = c.world == 1 && !c.no_eval_overlap && c.pots.size() > 1 && mode != MISTARK_EVAL_P;
    hipStream_t main_stream = c.stream;
    if (split) {
        if (!c.aux_stream) {
            MS_CHECK(hipStreamCreateWithFlags(&c.aux_stream, hipStreamNonBlocking));
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[0], hipEventDisableTiming));
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[1], hipEventDisableTiming));
        }
        MS_CHECK(hipEventRecord(c.aux_ev[0], main_stream));  // (zero fill of the gradient, uploads, the contact tables)
        MS_CHECK(hipStreamWaitEvent(c.aux_stream, c.aux_ev[0], 0));
        // the auxiliary stream's potentials add to their own copy of the gradient, folded in after the join: with one addition per row and
        // kernel (pooled potentials, single-node potentials) the sum of a row no longer depends on which stream got there first
        c.grad_aux.ensure((size_t)c.ndofs);
        fill_async(c.aux_stream, c.grad_aux.p, 0, (size_t)c.ndofs * sizeof(double));
    }
    double* const grad_main = c.grad.p;
    // The static part of the matrix can be gathered as soon as the element Hessians are there: on the auxiliary stream, behind its small
    // potentials, beside this stream's gradient gathers, reductions and the read-back the Newton loop takes its convergence decision from.
    // assemble() then waits for it and only adds the contact part. (Not for staged calls: their projection may run before the assembly.)
    // The event it waits for is recorded behind the LAST ELEMENT KERNEL of this stream, in front of the tets' gradient gathers (round 5: it used to
    // sit behind every gradient gather and the join, and 150-180 us of the 275 us gather ended up in front of the linear solve).
    c.static_assembled = false;
    const bool eager_asm = split && mode == MISTARK_EVAL_P_G_H && lazy && !c.atomic_assembly && !c.no_eager_assembly && !c.part[0].dirty && c.part[0].nnzb > 0;
    const bool early_asm = eager_asm && !c.late_eager_assembly;
    std::vector<Potential*> deferred_gathers;
    auto static_assembly = [&]() {
        if (!c.aux_ev[2]) {
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[2], hipEventDisableTiming));
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[3], hipEventDisableTiming));
        }
        MS_CHECK(hipEventRecord(c.aux_ev[3], main_stream));  // (every element kernel of this stream is in its queue; the auxiliary stream's own are in front of the gather)
        MS_CHECK(hipStreamWaitEvent(c.aux_stream, c.aux_ev[3], 0));
        c.have_hessians = true;
        c.stream = c.aux_stream;
        try {
            assemble_part(c, 0);
        } catch (...) {
            c.stream = main_stream;
            throw;
        }
        c.stream = main_stream;
        MS_CHECK(hipEventRecord(c.aux_ev[2], c.aux_stream));
        if (c.evt_armed[0]) evt_mark(c, 3, c.aux_stream);
        c.static_assembled = true;
    };
    try {
        // "small" = below EVAL_SMALL_POTENTIAL elements, or below a quarter of the largest potential (a million tets hide 172 k inertia nodes, too)
        int64_t n_max = 0;
        for (auto& P : c.pots) n_max = std::max<int64_t>(n_max, P.n_elem);
        const int64_t small = std::max<int64_t>(EVAL_SMALL_POTENTIAL, n_max / 4);
        // energy only: the small potentials share one launch (k_eval_p_multi)
        std::vector<MultiP> multi;
        MultiFirst mf;
        mf.n = 0;
        int multi_blocks = 0;
        const bool batch_p = mode == MISTARK_EVAL_P && c.world == 1 && !c.no_multi_eval_p && !c.kernel_dbg;
        // energy + gradient + Hessian: the contact and friction tables share one launch (k_eval_pgh_multi)
        std::vector<MultiPGH> multi_h;
        MultiFirst mfh;
        mfh.n = 0;
        int multi_h_blocks = 0;
        const bool batch_h = mode == MISTARK_EVAL_P_G_H && c.world == 1 && !c.no_multi_eval_pgh && !c.kernel_dbg;
        hipStream_t multi_h_stream = main_stream;
        double* multi_h_grad = grad_main;
        for (auto& P : c.pots) {
            int np = 0;
            if (batch_h && P.dyn_pool && P.n_elem < EVAL_SMALL_POTENTIAL && mfh.n < MULTI_P_MAX && joins_multi_pgh(c, P, &np)) {
                if (P.args.e_count == 0) continue;
                MultiPGH m;
                std::memset(&m, 0, sizeof(m));
                m.a = P.args;
                m.E = c.elemE.p + P.e_off;
                m.H = c.elemH.p + P.h_off;
                m.kind = P.kind;
                multi_h.push_back(m);
                mfh.b[mfh.n++] = multi_h_blocks;
                multi_h_blocks += grid_for((int64_t)P.args.e_count * np);
                const bool aux = split && P.n_elem < small;
                multi_h_stream = aux ? c.aux_stream : main_stream;
                multi_h_grad = aux ? c.grad_aux.p : grad_main;
                continue;
            }
            if (batch_p && P.kind != KIND_CUSTOM && P.kind >= 0 && P.n_elem < EVAL_SMALL_POTENTIAL && mf.n < MULTI_P_MAX) {
                if (P.args.e_count == 0) continue;
                MultiP m;
                std::memset(&m, 0, sizeof(m));
                m.a = P.args;
                m.E = c.elemE.p + P.e_off;
                m.kind = P.kind;
                multi.push_back(m);
                mf.b[mf.n++] = multi_blocks;
                multi_blocks += grid_for(P.args.e_count);
                continue;
            }
            const bool aux = split && P.n_elem < small;
            c.stream = aux ? c.aux_stream : main_stream;
            c.grad.p = aux ? c.grad_aux.p : grad_main;
            if (pre_pending)
                for (const Context::EvalPre::Item& it : pre.items)
                    if (&c.pots[(size_t)it.pot] == &P) {
                        MS_CHECK(hipStreamWaitEvent(c.stream, pre.ev_out, 0));
                        break;
                    }
            if (pre_ok && !aux) {  // evaluated ahead (eval_prelaunch) with the arguments it has now: only its gradient gather is left
                bool taken = false;
                for (const Context::EvalPre::Item& it : pre.items)
                    if (&c.pots[(size_t)it.pot] == &P && std::memcmp(&it.args, &P.args, sizeof(PotArgs)) == 0 &&
                        it.E == (const void*)((mode == MISTARK_EVAL_P ? c.elemE.p : c.elemE_pre.p) + P.e_off) &&
                        it.H == (mode != MISTARK_EVAL_P_G_H ? nullptr : (c.lazy_active ? (const void*)(c.elemHf.p + P.hf_off) : (const void*)(c.elemH.p + P.h_off)))) {
                        if (mode != MISTARK_EVAL_P)  // the energies it wrote aside (the line search's energy evaluation summed elemE meanwhile)
                            copy_async(c.stream, c.elemE.p + P.e_off, c.elemE_pre.p + P.e_off, (size_t)P.args.e_count * sizeof(double));
                        if (mode == MISTARK_EVAL_P) {
                        } else if (early_asm) deferred_gathers.push_back(&P);
                        else if (P.name == E_TetStrain::name) launch_tet_closed<E_TetStrain, true>(c, P, mode, false, true);
                        else launch_tet_closed<E_TetStrainEO, false>(c, P, mode, false, true);
                        taken = true;
                        c.n_prelaunch_used++;
                    }
                if (taken) continue;
            }
            if (early_asm && !aux && !c.force_generic && P.kind != KIND_CUSTOM && (P.name == E_TetStrain::name || P.name == E_TetStrainEO::name)) {
                // the kernel now, its gradient gather behind the event the static assembly waits for (below)
                if (P.name == E_TetStrain::name) launch_tet_closed<E_TetStrain, true>(c, P, mode, true);
                else launch_tet_closed<E_TetStrainEO, false>(c, P, mode, true);
                deferred_gathers.push_back(&P);
                continue;
            }
            launch_eval_kind(c, P, mode);
        }
        if (mfh.n > 0) {
            mfh.b[mfh.n] = multi_h_blocks;
            const size_t bytes = multi_h.size() * sizeof(MultiPGH);
            c.multi_h_dev.ensure(bytes);
            if (c.multi_h_sent.size() != bytes || std::memcmp(c.multi_h_sent.data(), multi_h.data(), bytes) != 0) {
                c.stream = multi_h_stream;
                h2d_small(c, c.multi_h_dev.p, multi_h.data(), bytes);  // (through a pinned slot, on the stream of the launch below)
                c.multi_h_sent.assign((const char*)multi_h.data(), (const char*)multi_h.data() + bytes);
            }
            hipLaunchKernelGGL(k_eval_pgh_multi, dim3(multi_h_blocks), dim3(BLOCK), 0, multi_h_stream, (const MultiPGH*)c.multi_h_dev.p, mfh, multi_h_grad);
            c.n_multi_pgh++;
        }
        if (mf.n > 0) {
            c.stream = main_stream;
            mf.b[mf.n] = multi_blocks;
            const size_t bytes = multi.size() * sizeof(MultiP);
            c.multi_p_dev.ensure(bytes);
            // (the descriptors change whenever a contact table does: sent when they differ from what the device holds)
            if (c.multi_p_sent.size() != bytes || std::memcmp(c.multi_p_sent.data(), multi.data(), bytes) != 0) {
                c.multi_p_sent.assign((const char*)multi.data(), (const char*)multi.data() + bytes);
                MS_CHECK(hipMemcpyAsync(c.multi_p_dev.p, c.multi_p_sent.data(), bytes, hipMemcpyHostToDevice, c.stream));
            }
            hipLaunchKernelGGL(k_eval_p_multi, dim3(multi_blocks), dim3(BLOCK), 0, c.stream, (const MultiP*)c.multi_p_dev.p, mf);
        }
    } catch (...) {
        c.stream = main_stream;
        c.grad.p = grad_main;
        throw;
    }
    c.stream = main_stream;
    c.grad.p = grad_main;
    if (split) MS_CHECK(hipEventRecord(c.aux_ev[1], c.aux_stream));  // (the join below waits for the small potentials, not for the gather queued behind them)
    if (split && mode == MISTARK_EVAL_P_G_H && c.evt_armed[0]) evt_mark(c, 2, c.aux_stream);
    if (early_asm) {
        static_assembly();
        for (Potential* P : deferred_gathers) {
            if (P->name == E_TetStrain::name) launch_tet_closed<E_TetStrain, true>(c, *P, mode, false, true);
            else launch_tet_closed<E_TetStrainEO, false>(c, *P, mode, false, true);
        }
    }
    if (split) {
        MS_CHECK(hipStreamWaitEvent(main_stream, c.aux_ev[1], 0));
        vec_axpby(c, c.grad.p, 1.0, c.grad.p, 1.0, c.grad_aux.p, c.ndofs);
    }
    // the device-resident tables' node gradients (contact, friction), row by row in sorted order: behind everything else, one addition per row
    if (mode != MISTARK_EVAL_P && !(c.kernel_dbg & 1)) dyn_grad_gather(c, c.grad.p);
    if (eager_asm && !early_asm) static_assembly();  // (option late_eager_assembly: where it sat through round 4, for A/B runs)
    // (the contact part's pattern ends in a read-back of its counts, for which the host waits: the energy / residual reductions of THIS stream
    // are queued first where they do not depend on it, so that they run while the host waits — see with_max below)
    bool pattern_pending = overlap_pattern;
    auto run_pattern = [&]() {
        if (!pattern_pending) return;
        pattern_pending = false;
        MS_CHECK(hipStreamWaitEvent(c.side_stream, c.side_ev[0], 0));
        hipStream_t main_stream = c.stream;
        c.stream = c.side_stream;  // (everything build_pattern launches and reads back goes through c.stream)
        try {
            build_pattern_part(c, 1);
        } catch (...) {
            c.stream = main_stream;
            throw;
        }
        c.stream = main_stream;
        MS_CHECK(hipEventRecord(c.side_ev[1], c.side_stream));
        if (c.evt_armed[0]) evt_mark(c, 5, c.side_stream);
        MS_CHECK(hipStreamWaitEvent(c.stream, c.side_ev[1], 0));
    };
    const bool with_max_early = grad_max_abs && mode != MISTARK_EVAL_P && c.world == 1 && c.n_elem_total > 0;
    if (!with_max_early) run_pattern();
    if (mode != MISTARK_EVAL_P && c.n_hot > 0)
        hipLaunchKernelGGL(k_fold_hot, dim3(grid_for(3 * (int64_t)c.n_hot)), dim3(BLOCK), 0, c.stream, (const double*)c.grad_hot.p, (const int32_t*)c.hot_rows.p, c.n_hot, c.grad.p);
    if (mode == MISTARK_EVAL_P_G_H) {
        fill_async(c.stream, c.is_projected.p, 0, (c.n_elem_total + 3) & ~(size_t)3);
        c.have_hessians = true;
        c.matrix_current = false;
        c.n_projected_total = 0;
    }
    double e = 0.0;
    const bool with_max = with_max_early;
    if (with_max) {  // energy and ||grad||_inf in one read-back
        const int g1 = grid_for((int64_t)c.n_elem_total, BLOCK, VEC_GRID), g2 = grid_for(c.ndofs, BLOCK, VEC_GRID);
        hipLaunchKernelGGL(k_sum, dim3(g1), dim3(BLOCK), 0, c.stream, (const double*)c.elemE.p, (int64_t)c.n_elem_total, c.partials.p);
        hipLaunchKernelGGL(k_max_abs, dim3(g2), dim3(BLOCK), 0, c.stream, (const double*)c.grad.p, c.ndofs, c.partials.p + g1);
        double* h = host_scratch(c, 2 * MAX_PARTIALS);
        // the partial sums leave the device BEFORE this stream is made to wait for the pattern chain (the publish kernel is queued behind the
        // reductions; the host picks the numbers up after the pattern's own read-back)
        if (mode == MISTARK_EVAL_P_G_H && c.evt_armed[0]) evt_mark(c, 4, c.stream);
        const bool published = fetch_partials_begin(c, g1 + g2, c.partials.p);
        const double t_issued = host_now_s();
        run_pattern();
        fetch_partials_end(c, g1 + g2, h, c.partials.p, published);
        if (mode == MISTARK_EVAL_P_G_H) {
            c.t_eval_issue += t_issued - t_enter;
            c.t_eval_wait += host_now_s() - t_issued;
        }
        double m = 0.0;
        for (int i = 0; i < g1; i++) e += h[i];
        for (int i = 0; i < g2; i++) m = std::max(m, h[g1 + i]);
        *grad_max_abs = m;
    } else if (c.world == 1) {
        e = c.n_elem_total ? reduce_sum(c, c.elemE.p, (int64_t)c.n_elem_total) : 0.0;
        if (grad_max_abs && mode != MISTARK_EVAL_P) *grad_max_abs = reduce_max_abs(c, c.grad.p, c.ndofs);
    } else {
        // sharded: this rank's share of the energy and the largest gradient entry on ITS rows, all-gathered and reduced in rank order;
        // then the gradient rows of the ghosts from their owners (the projection selects elements by the gradient of ALL their rows)
        const bool wg = mode != MISTARK_EVAL_P;
        const int g1 = grid_for((int64_t)std::max<size_t>(c.n_elem_total, 1), BLOCK, VEC_GRID), g2 = wg ? grid_for(3 * c.sh.n_own, BLOCK, VEC_GRID) : 0;
        hipLaunchKernelGGL(k_sum, dim3(g1), dim3(BLOCK), 0, c.stream, (const double*)c.elemE.p, (int64_t)c.n_elem_total, c.partials.p);
        if (wg) hipLaunchKernelGGL(k_max_abs_rows, dim3(g2), dim3(BLOCK), 0, c.stream, (const double*)c.grad.p, (const int32_t*)c.sh.grow.p, c.sh.n_own, c.partials.p + g1);
        double* h = host_scratch(c, 2 * MAX_PARTIALS);
        fetch_partials(c, g1 + g2, h, c.partials.p);
        double mine[2] = {0.0, 0.0}, all[2 * 64];
        for (int i = 0; i < g1; i++) mine[0] += h[i];
        for (int i = 0; i < g2; i++) mine[1] = std::max(mine[1], h[g1 + i]);
        shard_allgather_scalars(c, mine, 2, all);
        double m = 0.0;
        for (int r = 0; r < c.world; r++) {
            e += all[2 * r];
            m = std::max(m, all[2 * r + 1]);
        }
        if (grad_max_abs && wg) *grad_max_abs = m;
        if (wg) shard_halo_global(c, c.grad.p);
    }
    if (E) *E = e;
    if (grad_host && mode != MISTARK_EVAL_P) {
        const double* src = c.grad.p;
        if (c.world > 1) {  // the whole gradient for the caller: every rank contributes its rows
            shard_to_local(c, c.grad.p, c.q.p, false);
            shard_gather_global(c, c.q.p, c.z.p);
            src = c.z.p;
        }
        MS_CHECK(hipMemcpyAsync(grad_host, src, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
}

}  // namespace mistark
