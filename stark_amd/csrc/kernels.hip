// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the per-Newton-step hot path and their launchers.
//
//   element evaluation   : lane-per-(i,j) hyper-dual evaluation of the energy expressions (energies.hpp)
//   pattern build        : (block row, block col) keys -> radix sort -> unique -> 64-block tiles; the static part re-laid in
//                          row-aligned chunks of 8 tiles (build_aligned)
//   assembly             : element 3x3 blocks -> float BSR tiles (deterministic gather; float atomics as an option), block-Jacobi inverse
//   SpMV                 : one wavefront per chunk of complete rows, coalesced 16-B loads, in-wave segmented reduction
//   PCG                  : 3 kernels per iteration, device-resident convergence control, look-ahead batches
//   PSD projection       : per-element cyclic Jacobi eigen-decomposition; sharded runs exchange the matrix deltas
#include <atomic>
#include <chrono>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "dist.hpp"
#include "ipc_dev.hpp"
#include "engine.hpp"
#include "registry.hpp"
#include "tet_closed.hpp"
#include "tri_closed.hpp"
#include "contact_closed.hpp"

namespace mistark {

constexpr int BLOCK = 256;
constexpr int GRAD_LONG_ROW = 256;   // gradient incidences of a block row beyond which a wavefront sums the row (k_grad_gather_long)
constexpr int MAX_PARTIALS = 4096;   // max grid of any kernel that emits per-block partial sums
constexpr int VEC_GRID = 512;
// Jacobi sweeps stop when off(A)^2 <= tol * ||A||_F^2. Convergence is quadratic (a sweep squares off/||A||), so 1e-24 (off/||A|| <= 1e-12:
// eigenvalues and the rebuilt matrix to 1e-12 relative, three orders below the parity tolerance) saves the last sweep of 1e-30.
constexpr double JACOBI_OFF_TOL = 1e-24;
constexpr int64_t EVAL_SMALL_POTENTIAL = 32768;  // potentials with fewer elements are evaluated on the auxiliary stream (eval())
constexpr int PCG_GRID = 1024;  // vector kernels of the PCG (per-block partial sums: <= MAX_PARTIALS)

static inline int grid_for(int64_t n, int per_block = BLOCK, int cap = 1 << 30)
{
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

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

// element of a kernel's local index le, and its position in the pools (element energies, element Hessians)
__device__ __forceinline__ int elem_of(const PotArgs& a, int le) { return a.elem_list ? (int)a.elem_list[le] : a.e_begin + le; }
__device__ __forceinline__ int pool_of(const PotArgs& a, int le) { return a.elem_list ? le : a.e_begin + le; }
// sharded runs: an element on an interface is evaluated by every rank that owns one of its rows; its energy counts where the row of its
// first DoF block lives
__device__ __forceinline__ bool energy_here(const PotArgs& a, int e)
{
    if (!a.lrow) return true;
    const int l = a.lrow[a.dof_row_off[0] + a.conn[(size_t)e * a.conn_stride + a.dof_col[0]]];
    return l >= 0 && l < a.n_own;
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
__global__ __launch_bounds__(BLOCK) void k_eval_pgh(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    constexpr int NB = En::NB, n = 3 * NB, NP = n * (n + 1) / 2;
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
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
        if (a.hot_base[ba] >= 0) atomicAdd(&a.grad_hot[((size_t)(blockIdx.x & (HOT_WAYS - 1)) * a.n_hot + a.hot_base[ba] + node) * 3 + ii], r.a);
        else atomicAdd(&grad[3 * (size_t)(a.dof_row_off[ba] + node) + ii], r.a);
    }
    if (first) elemE[pe] = energy_here(a, e) ? r.v : 0.0;
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
struct TetBlockStagedSink
{
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
__host__ __device__ constexpr int tet_pair_index(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }  // a <= b: 0..9
struct TetBlockFloatSink
{
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
__global__ __launch_bounds__(BLOCK) void k_eval_tet_closed(PotArgs a, double* __restrict__ elemE, double* __restrict__ elemH, float* __restrict__ elemHf, double* __restrict__ grad)
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
    if (MODE == TET_PGH || MODE == TET_H_LIST) {
        TetBlockStagedSink sink{stage + wave * 9 * 64, elemH + (size_t)pe_wave * 9, (size_t)a.n_pool * 9, lane, min(64, a.e_count - le_wave)};
        tet_closed_eval_to<FULL>(in, E, g, sink, true);
    } else if (MODE == TET_PGH_F) {
        TetBlockFloatSink sink{reinterpret_cast<float*>(stage) + wave * 9 * 64, elemHf + (size_t)pe_wave * ((a.dbg & 4) ? 90 : 9), (size_t)a.n_pool * 9, lane, min(64, a.e_count - le_wave), a.dbg};
        tet_closed_eval_to<FULL>(in, E, g, sink, true);
    } else {
        tet_closed_eval<FULL>(in, E, g, nullptr, 0, false);
    }
    if (!valid || MODE == TET_H_LIST) return;
    elemE[pool_of(a, le)] = energy_here(a, e) ? E : 0.0;
    if (a.dbg & 1) return;  // measurement switch: no gradient output
    if (a.gpool) {  // node gradients to the pool, summed per block row by k_grad_gather (no atomics)
        const int pe = pool_of(a, le);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double* gp = a.gpool + ((size_t)k * a.n_gpool + pe) * 3;
            gp[0] = g[3 * k];
            gp[1] = g[3 * k + 1];
            gp[2] = g[3 * k + 2];
        }
        return;
    }
    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t row = (size_t)(a.dof_row_off[k] + ce[a.dof_col[k]]);
        atomicAdd(&grad[3 * row], g[3 * k]);
        atomicAdd(&grad[3 * row + 1], g[3 * k + 1]);
        atomicAdd(&grad[3 * row + 2], g[3 * k + 2]);
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
static void launch_tet_closed_list(Context& c, Potential& P, const uint32_t* list, int n_list, double* H, int n_pool)
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
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_down(v, d, 64));
    return v;
}
// Sum over the 256 threads of a block; result valid in every thread. Deterministic.
__device__ __forceinline__ double block_sum(double v, double* sm /*[4]*/)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ double block_max(double v, double* sm)
{
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
// Deterministic sum of `n` per-block partials, computed redundantly by every block that needs the scalar.
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n, double* sm, int stride = 1)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) s += part[(size_t)i * stride];
    return block_sum(s, sm);
}

// two sums at once (one pair of barriers)
__device__ __forceinline__ void sum_partials2(const double* __restrict__ pa, const double* __restrict__ pb, int n, double* sm /*[8]*/, int stride, double& a, double& b)
{
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        sa += pa[(size_t)i * stride];
        sb += pb[(size_t)i * stride];
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sm[w] = sa;
        sm[4 + w] = sb;
    }
    __syncthreads();
    a = sm[0] + sm[1] + sm[2] + sm[3];
    b = sm[4] + sm[5] + sm[6] + sm[7];
}
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

static double* host_scratch(Context& c, size_t n)
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
void h2d_staged(Context& c, void* dst_dev, const void* src_host, size_t bytes)
{
    constexpr size_t CHUNK = (size_t)4 << 20;
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
constexpr uint32_t NO_SRC = 0xFFFFFFFFu;
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

// Where the gather assembly reads a contribution from: one descriptor per sorted key. Bit 31: float pool (elemHf) instead of the double pool
// (elemH), bit 30: read the stored block transposed (lazy potentials keep the upper block triangle only), bits 0..29: 3x3 block index in that
// pool. NO_SRC: no data (structural diagonal keys; multi-GPU: elements of other ranks, the sum over ranks restores them).
constexpr uint32_t DESC_FLOAT = 0x80000000u, DESC_TRANS = 0x40000000u, DESC_MASK = 0x3fffffffu;
struct DescRange  // keys [kp_off, kp_off + nn * n_elem) of one potential
{
    uint32_t kp_off, n_elem, NB, e_begin, e_count;
    uint32_t pool_blk;   // first block of the potential in its pool
    uint32_t n_pool;     // pool stride (elements per block pair)
    uint32_t lazy;       // float pool, upper block triangle (tet_pair_index)
};
__device__ __forceinline__ uint32_t make_desc(uint32_t kp, const DescRange* __restrict__ rg, int n_rg);
__global__ __launch_bounds__(BLOCK) void k_make_desc(const uint32_t* __restrict__ sidx, size_t n, const DescRange* __restrict__ rg, int n_rg, uint32_t* __restrict__ desc)
{
    const size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) desc[k] = make_desc(sidx[k], rg, n_rg);
}
constexpr int DESC_TABLE_MAX = 64;
struct DescTable
{
    DescRange r[DESC_TABLE_MAX];
    int n;
};
__global__ __launch_bounds__(BLOCK) void k_make_desc_tab(const uint32_t* __restrict__ sidx, size_t n, DescTable tab, uint32_t* __restrict__ desc)
{
    __shared__ DescRange rg[DESC_TABLE_MAX];
    for (int i = threadIdx.x; i < tab.n; i += BLOCK) rg[i] = tab.r[i];
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) desc[k] = make_desc(sidx[k], rg, tab.n);
}
__device__ __forceinline__ uint32_t make_desc(uint32_t kp, const DescRange* __restrict__ rg, int n_rg)
{
    uint32_t d = NO_SRC;
    if (kp != NO_SRC) {
        int lo = 0, hi = n_rg - 1;  // last range with kp_off <= kp
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rg[mid].kp_off <= kp) lo = mid;
            else hi = mid - 1;
        }
        const DescRange r = rg[lo];
        const uint32_t off = kp - r.kp_off, ab = off / r.n_elem, e = off - ab * r.n_elem;
        if (e >= r.e_begin && e < r.e_begin + r.e_count) {
            if (r.lazy) {
                const uint32_t a = ab / r.NB, b = ab - a * r.NB;
                const uint32_t pr = (uint32_t)(a > b ? tet_pair_index((int)b, (int)a) : tet_pair_index((int)a, (int)b));
                d = DESC_FLOAT | (a > b ? DESC_TRANS : 0u) | (r.lazy == 2u ? r.pool_blk + e * 10u + pr : r.pool_blk + pr * r.n_pool + e);  // (2: element-major pool)
            } else {
                d = r.pool_blk + ab * r.n_pool + e;
            }
        }
    }
    return d;
}
constexpr int CHUNK_BLOCKS = 256;
constexpr int DYN_SHORT_ROW = 32;  // contact rows of a node hold a handful of blocks; only the rows of rigid bodies in contact are long
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
constexpr uint32_t LONG_SLOT = 48;  // BSR blocks with more contributions than this are summed by a whole wavefront (k_assemble_long)
constexpr uint32_t VERY_LONG_SLOT = 4096;  // ... and beyond this by VLONG_SPLIT wavefronts and a second pass (the blocks of a rigid body under 10^4..10^5 contacts)
constexpr int VLONG_SPLIT = 64;
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
            MS_CHECK(hipMemcpyAsync(c.dyn_desc.p, dyn_desc.data(), dyn_desc.size() * sizeof(DynIncDesc), hipMemcpyHostToDevice, c.stream));
            MS_CHECK(hipStreamSynchronize(c.stream));  // (host vector is a temporary)
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
static void assemble_part(Context& c, int part);
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
    pre.valid = true;
    pre.mode = mode;
    pre.lazy_active = lazy_active;
}
void eval(Context& c, int mode, double* E, double* grad_host, double* grad_max_abs, bool lazy)
{
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
        for (auto& P : c.pots) {
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
                        } else if (P.name == E_TetStrain::name) launch_tet_closed<E_TetStrain, true>(c, P, mode, false, true);
                        else launch_tet_closed<E_TetStrainEO, false>(c, P, mode, false, true);
                        taken = true;
                        c.n_prelaunch_used++;
                    }
                if (taken) continue;
            }
            launch_eval_kind(c, P, mode);
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
    if (split) {
        MS_CHECK(hipEventRecord(c.aux_ev[1], c.aux_stream));
        MS_CHECK(hipStreamWaitEvent(main_stream, c.aux_ev[1], 0));
        vec_axpby(c, c.grad.p, 1.0, c.grad.p, 1.0, c.grad_aux.p, c.ndofs);
    }
    // the device-resident tables' node gradients (contact, friction), row by row in sorted order: behind everything else, one addition per row
    if (mode != MISTARK_EVAL_P && !(c.kernel_dbg & 1)) dyn_grad_gather(c, c.grad.p);
    // The static part of the matrix can be gathered as soon as the element Hessians are there: on the auxiliary stream (idle by now),
    // beside this stream's gradient gather, reductions and the read-back the Newton loop takes its convergence decision from. assemble()
    // then waits for it and only adds the contact part. (Not for staged calls: their projection may run before the assembly.)
    c.static_assembled = false;
    if (split && mode == MISTARK_EVAL_P_G_H && lazy && !c.atomic_assembly && !c.no_eager_assembly && !c.part[0].dirty && c.part[0].nnzb > 0) {
        if (!c.aux_ev[2]) {
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[2], hipEventDisableTiming));
            MS_CHECK(hipEventCreateWithFlags(&c.aux_ev[3], hipEventDisableTiming));
        }
        MS_CHECK(hipEventRecord(c.aux_ev[3], main_stream));  // (every element kernel is in this stream's queue, or joined into it)
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
        c.static_assembled = true;
    }
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
        const bool published = fetch_partials_begin(c, g1 + g2, c.partials.p);
        run_pattern();
        fetch_partials_end(c, g1 + g2, h, c.partials.p, published);
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

// ======================================================================================================================
// PSD projection (project_to_PD.cpp:12-32; ElementHessians.cpp:48-67,79-182).
//   k_project_select_multi : marks the not-yet-projected elements that touch an active block row and appends them to a list
//   k_project_eig    : one WAVEFRONT per listed element: parallel-order cyclic Jacobi on the n x n matrix held in LDS
//                      (n/2 disjoint rotations per round, lanes own matrix entries), eigenvalues < eps clamped (or mirrored),
//                      V L V^T rebuilt only if something changed; the difference (projected - original) is added to the
//                      already assembled float BSR (the reference's update_global, ElementHessians.cpp:258-294).
// ======================================================================================================================
// position of component `comp` (row-major 3x3) of BSR block `slot` inside the 64-block tile layout (see SpMV)
__device__ __forceinline__ size_t tile_val_index(uint32_t slot, int comp)
{
    const size_t base = (size_t)(slot >> 6) * 576;
    const uint32_t lane = slot & 63u;
    if (comp < 4) return base + lane * 4 + comp;
    if (comp < 8) return base + 256 + lane * 4 + (comp - 4);
    return base + 512 + lane;
}

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

static void gather_part(Context& c, int part, const uint8_t* only_dirty);
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

// ======================================================================================================================
// Assembly: element 3x3 blocks -> float BSR tiles
// ======================================================================================================================
__global__ __launch_bounds__(BLOCK) void k_assemble(const double* __restrict__ elemH, int64_t n_blocks_total, const uint32_t* __restrict__ slot_of_src, float* __restrict__ vals)
{
    // one lane per (element block, component): 9 consecutive lanes read 72 contiguous bytes
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_blocks_total * 9) return;
    const int64_t blk = t / 9;
    const int comp = (int)(t - blk * 9);
    const uint32_t slot = slot_of_src[blk];
    atomicAdd(&vals[tile_val_index(slot, comp)], (float)elemH[t]);
}
// closed-form inverse of a SYMMETRIC 3x3 in float, reciprocal of the determinant through double (BlockedSparseMatrix.h:1198-1214)
__device__ __forceinline__ void sym3_inverse(const float* m, float* o)
{
    const float tmp0 = m[4] * m[8];
    const float tmp1 = m[5] * m[5];
    const float tmp2 = m[2] * m[5];
    const float tmp3 = m[1] * m[1];
    const float tmp4 = m[2] * m[2];
    const float det = m[0] * tmp0 - m[0] * tmp1 + 2 * m[1] * tmp2 - m[4] * tmp4 - m[8] * tmp3;
    const float tmp5 = (float)(1.0 / (double)det);
    o[8] = tmp5 * (m[0] * m[4] - tmp3);
    o[4] = tmp5 * (m[0] * m[8] - tmp4);
    o[0] = tmp5 * (tmp0 - tmp1);
    o[3] = -tmp5 * (m[1] * m[8] - tmp2);
    o[1] = o[3];
    o[6] = tmp5 * (m[1] * m[5] - m[4] * m[2]);
    o[2] = o[6];
    o[7] = -tmp5 * (m[0] * m[5] - m[1] * m[2]);
    o[5] = o[7];
}
__global__ __launch_bounds__(BLOCK) void k_block_diag_inverse(const float* __restrict__ vals, const int32_t* __restrict__ diag_slot, const float* __restrict__ vals_dyn,
                                                              const int32_t* __restrict__ diag_slot_dyn, int64_t nbr, float* __restrict__ dinv)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= nbr) return;
    const uint32_t s = (uint32_t)diag_slot[r];
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = vals[tile_val_index(s, k)];
    if (vals_dyn) {
        const int32_t sd = diag_slot_dyn[r];
        if (sd >= 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) m[k] += vals_dyn[tile_val_index((uint32_t)sd, k)];
        }
    }
    sym3_inverse(m, dinv + 9 * r);
}

// Gather assembly (default): the contributions of a BSR block are summed in the fixed order of the sorted pattern keys: no atomics,
// deterministic, double accumulation rounded once to float (k_assemble_gather; blocks with many contributions: k_assemble_long / _vlong).
// contribution `desc` (k_make_desc), component comp (row-major) of the 3x3 block
// Branch-free on purpose: the callers keep eight of these in flight per lane, and loads under divergent control flow are issued one after
// the other (measured: 630 instead of 460 us for the 1M-tet matrix). Both pools are read, the one that does not apply at its first element.
__device__ __forceinline__ double contrib(const double* __restrict__ elemH, const float* __restrict__ elemHf, uint32_t desc, int comp, int comp_t)
{
    const bool none = desc == NO_SRC;  // (the structural diagonal keys carry no data)
    const bool f = !none && (desc & DESC_FLOAT) != 0u;
    const bool d = !none && !f;
    const size_t blk = (size_t)(desc & DESC_MASK) * 9;
    const float vf = elemHf[f ? blk + (size_t)((desc & DESC_TRANS) ? comp_t : comp) : (size_t)0];
    const double vd = elemH[d ? (size_t)desc * 9 + (size_t)comp : (size_t)0];
    return f ? (double)vf : (d ? vd : 0.0);
}
// one wavefront per long block (e.g. the diagonal block of a rigid body touched by thousands of contacts): lanes take
// contributions k0 + lane, k0 + lane + 64, ... and the nine sums are reduced across the wave; the order is fixed by the sorted keys
__global__ __launch_bounds__(BLOCK) void k_assemble_long(const double* __restrict__ elemH, const float* __restrict__ elemHf, const uint32_t* __restrict__ slot_start,
                                                        const uint32_t* __restrict__ sorted_src, const uint32_t* __restrict__ list, int n_long, const uint32_t* __restrict__ store_slot,
                                                        float* __restrict__ vals, const uint8_t* __restrict__ only_dirty)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_long) return;
    const int lane = threadIdx.x & 63;
    const uint32_t slot = list[w];
    if (only_dirty && !only_dirty[store_slot ? store_slot[slot] : slot]) return;  // (flags are indexed like the values: by storage position)
    const uint32_t k0 = slot_start[slot], k1 = slot_start[slot + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = k0 + lane; k < k1; k += 64) {
        const uint32_t src = sorted_src[k];
        if (src == NO_SRC) continue;
#pragma unroll
        for (int c = 0; c < 9; c++) acc[c] += contrib(elemH, elemHf, src, c, (c % 3) * 3 + c / 3);
    }
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const double v = wave_sum(acc[c]);
        if (lane == 0) vals[tile_val_index(store_slot ? store_slot[slot] : slot, c)] = (float)v;
    }
}
// very long blocks: VLONG_SPLIT wavefronts per block sum contiguous ranges of its contribution list, a second pass adds the partial sums in
// range order (deterministic like the one-wavefront version, 64 times the parallelism: 2.2 ms -> tens of us for the four diagonal blocks
// of a floor under 136 k contact and friction rows)
__global__ __launch_bounds__(BLOCK) void k_assemble_vlong_part(const double* __restrict__ elemH, const float* __restrict__ elemHf, const uint32_t* __restrict__ slot_start,
                                                              const uint32_t* __restrict__ sorted_src, const uint32_t* __restrict__ list, int n_vlong, double* __restrict__ part,
                                                              const uint8_t* __restrict__ only_dirty, const uint32_t* __restrict__ store_slot)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_vlong * VLONG_SPLIT) return;
    const int lane = threadIdx.x & 63;
    const int b = w / VLONG_SPLIT, j = w - b * VLONG_SPLIT;
    const uint32_t slot = list[b];
    if (only_dirty && !only_dirty[store_slot ? store_slot[slot] : slot]) return;
    const uint32_t k0 = slot_start[slot], k1 = slot_start[slot + 1];
    const uint32_t chunk = (k1 - k0 + VLONG_SPLIT - 1) / VLONG_SPLIT;
    const uint32_t c0 = k0 + (uint32_t)j * chunk, c1 = min(k1, c0 + chunk);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = c0 + lane; k < c1; k += 64) {
        const uint32_t src = sorted_src[k];
        if (src == NO_SRC) continue;
#pragma unroll
        for (int c = 0; c < 9; c++) acc[c] += contrib(elemH, elemHf, src, c, (c % 3) * 3 + c / 3);
    }
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const double v = wave_sum(acc[c]);
        if (lane == 0) part[(size_t)w * 9 + c] = v;
    }
}
__global__ __launch_bounds__(BLOCK) void k_assemble_vlong_fold(const double* __restrict__ part, const uint32_t* __restrict__ list, int n_vlong, const uint32_t* __restrict__ store_slot,
                                                              float* __restrict__ vals, const uint8_t* __restrict__ only_dirty)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_vlong) return;
    const int lane = threadIdx.x & 63;
    static_assert(VLONG_SPLIT == 64, "one lane per partial sum");
    const uint32_t slot = list[b];
    if (only_dirty && !only_dirty[store_slot ? store_slot[slot] : slot]) return;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const double v = wave_sum(part[((size_t)b * VLONG_SPLIT + lane) * 9 + c]);
        if (lane == 0) vals[tile_val_index(store_slot ? store_slot[slot] : slot, c)] = (float)v;
    }
}
// One lane per BSR block: nine double accumulators, the contributions of the block summed in list order (deterministic, one float
// rounding at the end), four contributions in flight per lane. A float contribution is 36 contiguous bytes (three 12-byte loads), a
// double one 72; consecutive lanes own consecutive blocks, whose contributions come from neighbouring elements, and write neighbouring
// float4s of the tile layout. (The earlier lane-per-(block, component) version was bound by the latency of its dependent loads:
// 360 k wavefronts with three round trips each, 460 us for the 1M-tet matrix; its float-pool variant 630-770 us.)
struct F3
{
    float x, y, z;
};
__global__ __launch_bounds__(BLOCK) void k_assemble_gather(const double* __restrict__ elemH, const float* __restrict__ elemHf, const uint32_t* __restrict__ slot_start,
                                                           const uint32_t* __restrict__ sorted_src, int64_t nnzb, const uint32_t* __restrict__ store_slot, float* __restrict__ vals,
                                                           const uint8_t* __restrict__ only_dirty)
{
    const int64_t slot = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= nnzb) return;
    if (only_dirty && !only_dirty[store_slot ? store_slot[slot] : (uint32_t)slot]) return;  // (project(): only the blocks a projection round touched; flags by storage position)
    const uint32_t k0 = slot_start[slot], k1 = slot_start[slot + 1];
    if (k1 - k0 > LONG_SLOT) return;  // k_assemble_long
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t kb = k0; kb < k1; kb += 4) {
        uint32_t d[4];
        F3 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) d[u] = kb + u < k1 ? sorted_src[kb + u] : NO_SRC;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool f = d[u] != NO_SRC && (d[u] & DESC_FLOAT) != 0u;
            const F3* src = reinterpret_cast<const F3*>(elemHf + (f ? (size_t)(d[u] & DESC_MASK) * 9 : (size_t)0));  // (block 0 of the pool when not a float contribution)
            v[u][0] = src[0];
            v[u][1] = src[1];
            v[u][2] = src[2];
        }
        bool any_double = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool f = d[u] != NO_SRC && (d[u] & DESC_FLOAT) != 0u;
            const bool t = (d[u] & DESC_TRANS) != 0u;
            any_double = any_double || (d[u] != NO_SRC && !f);
            if (f) {
                acc[0] += (double)v[u][0].x;
                acc[1] += (double)(t ? v[u][1].x : v[u][0].y);
                acc[2] += (double)(t ? v[u][2].x : v[u][0].z);
                acc[3] += (double)(t ? v[u][0].y : v[u][1].x);
                acc[4] += (double)v[u][1].y;
                acc[5] += (double)(t ? v[u][2].y : v[u][1].z);
                acc[6] += (double)(t ? v[u][0].z : v[u][2].x);
                acc[7] += (double)(t ? v[u][1].z : v[u][2].y);
                acc[8] += (double)v[u][2].z;
            }
        }
        if (any_double) {  // contributions from the double pool (potentials off the lazy path; every potential on staged calls)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (d[u] == NO_SRC || (d[u] & DESC_FLOAT)) continue;
                const double* h = elemH + (size_t)d[u] * 9;
#pragma unroll
                for (int c = 0; c < 9; c++) acc[c] += h[c];
            }
        }
    }
    const uint32_t pos = store_slot ? store_slot[slot] : (uint32_t)slot;
    float* tile = vals + (size_t)(pos >> 6) * 576;
    const uint32_t lane = pos & 63u;
    reinterpret_cast<float4*>(tile)[lane] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
    reinterpret_cast<float4*>(tile)[64 + lane] = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
    tile[512 + lane] = (float)acc[8];
}

// descriptors of the gather lists for the current state of the pools (lazy or not): once per pattern and lazy state
static void make_descriptors(Context& c, int part)
{
    BsrPart& m = c.part[part];
    if (m.desc_lazy == (c.lazy_active ? 1 : 0) || m.n_keys == 0) return;
    std::vector<DescRange> rg;
    for (auto& P : c.pots) {
        if (P.part != part || P.n_key == 0) continue;
        const bool lazy = c.lazy_active && P.lazy_capable;
        // (key space and pools hold the n_key elements this context evaluates: all of them, or the rank's list)
        rg.push_back(DescRange{(uint32_t)P.kp_off, (uint32_t)P.n_key, (uint32_t)P.NB, 0u, (uint32_t)P.n_key, lazy ? (uint32_t)(P.hf_off / 9) : (uint32_t)P.k_off,
                               lazy ? (uint32_t)P.n_pool_f : (uint32_t)P.n_key, lazy ? (c.hf_layout ? 2u : 1u) : 0u});
    }
    if (c.hess_total / 9 > DESC_MASK || c.hf_total / 9 > DESC_MASK) throw Error("element-Hessian pool too large for the gather descriptors");
    m.sorted_desc.ensure(m.n_keys);
    if (rg.size() <= (size_t)DESC_TABLE_MAX) {  // the table travels in the kernel arguments: no copy, no synchronisation
        DescTable tab{};
        tab.n = (int)rg.size();
        for (size_t i = 0; i < rg.size(); i++) tab.r[i] = rg[i];
        hipLaunchKernelGGL(k_make_desc_tab, dim3(grid_for((int64_t)m.n_keys)), dim3(BLOCK), 0, c.stream, m.sorted_src, m.n_keys, tab, m.sorted_desc.p);
    } else {
        c.src_ranges.ensure(rg.size() * sizeof(DescRange));
        MS_CHECK(hipMemcpyAsync(c.src_ranges.p, rg.data(), rg.size() * sizeof(DescRange), hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(k_make_desc, dim3(grid_for((int64_t)m.n_keys)), dim3(BLOCK), 0, c.stream, m.sorted_src, m.n_keys, (const DescRange*)c.src_ranges.p, (int)rg.size(), m.sorted_desc.p);
        MS_CHECK(hipStreamSynchronize(c.stream));  // rg is a temporary
    }
    m.desc_lazy = c.lazy_active ? 1 : 0;
}
// the blocks of a matrix part summed from the pools in sorted-key order; only_dirty: just the flagged blocks (BsrPart::slot_dirty)
static void gather_part(Context& c, int part, const uint8_t* only_dirty)
{
    BsrPart& m = c.part[part];
    make_descriptors(c, part);
    const uint32_t* store = part == 0 ? m.store_slot.p : nullptr;
    const uint32_t* desc = m.sorted_desc.p;
    hipLaunchKernelGGL(k_assemble_gather, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.nnzb, store, m.vals.p, only_dirty);
    if (m.n_long > 0)
        hipLaunchKernelGGL(k_assemble_long, dim3((m.n_long + 3) / 4), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.long_slots.p, m.n_long, store, m.vals.p, only_dirty);
    if (m.n_vlong > 0) {
        c.vlong_part.ensure((size_t)m.n_vlong * VLONG_SPLIT * 9);
        hipLaunchKernelGGL(k_assemble_vlong_part, dim3((m.n_vlong * VLONG_SPLIT + 3) / 4), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.vlong_slots.p, m.n_vlong,
                           c.vlong_part.p, only_dirty, store);
        hipLaunchKernelGGL(k_assemble_vlong_fold, dim3((m.n_vlong + 3) / 4), dim3(BLOCK), 0, c.stream, (const double*)c.vlong_part.p, m.vlong_slots.p, m.n_vlong, store, m.vals.p, only_dirty);
    }
}
static void assemble_part(Context& c, int part)
{
    {
        BsrPart& m = c.part[part];
        if (m.nnzb == 0) return;
        if (c.atomic_assembly && c.world == 1) {
            MS_CHECK(hipMemsetAsync(m.vals.p, 0, (size_t)m.ntiles * 576 * sizeof(float), c.stream));
            for (auto& P : c.pots) {
                const int64_t nblk = (int64_t)P.n_elem * P.NB * P.NB;
                if (P.part != part || nblk == 0) continue;
                hipLaunchKernelGGL(k_assemble, dim3(grid_for(nblk * 9)), dim3(BLOCK), 0, c.stream, c.elemH.p + P.h_off, nblk, m.slot_of_src.p + P.kp_off, m.vals.p);
            }
        } else {
            gather_part(c, part, nullptr);
        }
        m.have_matrix = true;
    }
}
void assemble(Context& c)
{
    ensure_pattern(c);
    if (!c.have_hessians) throw Error("assemble: no element Hessians (call eval with MISTARK_EVAL_P_G_H first)");
    if (c.static_assembled) MS_CHECK(hipStreamWaitEvent(c.stream, c.aux_ev[2], 0));  // eval() gathered the static part on the auxiliary stream
    for (int part = 0; part < 2; part++) {
        if (part == 0 && c.static_assembled) continue;
        assemble_part(c, part);
    }
    c.static_assembled = false;
    c.have_matrix = true;
    c.matrix_current = true;
}
void build_preconditioner(Context& c)
{
    if (!c.have_matrix) throw Error("preconditioner: matrix not assembled");
    const BsrPart& d = c.part[1];
    const int64_t nr = c.mrows();
    if (nr > 0)
        hipLaunchKernelGGL(k_block_diag_inverse, dim3(grid_for(nr)), dim3(BLOCK), 0, c.stream, c.part[0].vals.p, c.diag_slot[0].p, d.nnzb ? d.vals.p : (const float*)nullptr,
                           c.diag_slot[1].p, nr, c.dinv.p);
}

// ======================================================================================================================
// SpMV  y = A x  (+ optional fused dot  pdot . y  -> per-block partials)
// Tiles of 64 consecutive 3x3 blocks (CSR order). Values are laid out per tile as
// float4 q0[64] | float4 q1[64] | float s[64] so that every load instruction of a wave is a fully coalesced
// 1 KiB (dwordx4) or 256 B (dword) request: 36 B per block, no padding. Column word: bit 31 marks the last block of a row.
// Rows are reduced inside the wave by a DPP segmented scan; rows that straddle tiles of a chunk are carried in registers
// (spmv_chunked_static). The static part's tiles are grouped in row-aligned chunks (build_aligned), one wavefront each.
// ======================================================================================================================
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr(double v)
{
    // v_mov_b32_dpp row_shr:n on both halves; lanes without a source inside their 16-lane row receive 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// y = A_static x (build_aligned). One wavefront per chunk of SPMV_CHUNK_TILES tiles; a chunk holds complete rows, so the wavefront
// neither reads a neighbour's tile nor hands a partial row on. Per tile: the column words and values (prefetched one tile ahead), the
// x gather, the nine float -> double conversions and FMAs of the reference (BlockedSparseMatrix.h:986-1138), then a segmented inclusive
// scan over the 64 lanes (DPP row shifts + three scalar carries, no LDS) whose row-end lanes write y; a row that continues into the next
// tile of the chunk is carried in registers. Every row is written exactly once: no atomics, no zero fill, deterministic.
// What the SpMV multiplies with. XPlain: a vector in memory. XDir: the PCG's search direction p = z + beta p_old computed on the fly, so that
// the direction update needs no kernel of its own (k_pcg_dir: one launch, one dependent-kernel boundary and 12 MB of vector traffic per
// iteration at 1M tets); the lane that finishes a row also stores p[row] for k_pcg_step and the next iteration. MEASURED (configs[3],
// profiles/r02_v2_fuse_dir_kernel_stats.txt): the second gathered vector costs the SpMV 8 us (23.9 -> 32), more than the 7.3 us kernel it
// replaces (1.40 instead of 1.32 ms per solve); identical iteration counts. Kept as option "fuse_dir" and as a cross-check of the solver.
struct XPlain
{
    const double* x;
    const double* pd;  // fused dot: sum of pd[row] . y[row] (nullptr: none)
    __device__ __forceinline__ void load(size_t c3, double& x0, double& x1, double& x2) const
    {
        x0 = x[c3];
        x1 = x[c3 + 1];
        x2 = x[c3 + 2];
    }
    __device__ __forceinline__ bool has_dot() const { return pd != nullptr; }
    __device__ __forceinline__ double row_dot(size_t r3, double y0, double y1, double y2) const { return pd[r3] * y0 + pd[r3 + 1] * y1 + pd[r3 + 2] * y2; }
    // the same in two halves: the row's entries of pd are loaded EARLY (with the tile's gathers), the product is formed after the row sums
    __device__ __forceinline__ void row_pre(size_t r3, double& p0, double& p1, double& p2) const
    {
        p0 = pd[r3];
        p1 = pd[r3 + 1];
        p2 = pd[r3 + 2];
    }
    __device__ __forceinline__ double row_dot_pre(size_t, double p0, double p1, double p2, double y0, double y1, double y2) const { return p0 * y0 + p1 * y1 + p2 * y2; }
};
struct XDir
{
    const double* z;
    const double* pold;
    double* pnew;
    double beta;
    __device__ __forceinline__ void load(size_t c3, double& x0, double& x1, double& x2) const
    {
        x0 = z[c3] + beta * pold[c3];
        x1 = z[c3 + 1] + beta * pold[c3 + 1];
        x2 = z[c3 + 2] + beta * pold[c3 + 2];
    }
    __device__ __forceinline__ bool has_dot() const { return true; }
    // the row's own entries of p: stored (every block row ends in exactly one lane of the static part), and p[row] . y[row] for p.Ap
    __device__ __forceinline__ double row_dot(size_t r3, double y0, double y1, double y2) const
    {
        double p0, p1, p2;
        load(r3, p0, p1, p2);
        pnew[r3] = p0;
        pnew[r3 + 1] = p1;
        pnew[r3 + 2] = p2;
        return p0 * y0 + p1 * y1 + p2 * y2;
    }
    __device__ __forceinline__ void row_pre(size_t r3, double& p0, double& p1, double& p2) const { load(r3, p0, p1, p2); }
    __device__ __forceinline__ double row_dot_pre(size_t r3, double p0, double p1, double p2, double y0, double y1, double y2) const
    {
        pnew[r3] = p0;
        pnew[r3 + 1] = p1;
        pnew[r3 + 2] = p2;
        return p0 * y0 + p1 * y1 + p2 * y2;
    }
    // (contact part: its rows are stored by the static part; here only the product is needed)
    __device__ __forceinline__ double row_dot_nostore(size_t r3, double y0, double y1, double y2) const
    {
        double p0, p1, p2;
        load(r3, p0, p1, p2);
        return p0 * y0 + p1 * y1 + p2 * y2;
    }
};
__device__ __forceinline__ double row_dot_nostore(const XPlain& X, size_t r3, double y0, double y1, double y2) { return X.row_dot(r3, y0, y1, y2); }
// Measurement only (spmv_variant 12; north_star names "SoA node/DoF arrays"): the vectors as three arrays x[n], y[n], z[n] instead of the
// reference's interleaved (x, y, z) per node. A block's gather then touches three cache lines in three regions instead of one 24-byte run.
struct XSoA
{
    const double* x;  // [3][n]: component-major copy of the vector
    size_t n;
    __device__ __forceinline__ void load(size_t c3, double& x0, double& x1, double& x2) const
    {
        const size_t c = c3 / 3;
        x0 = x[c];
        x1 = x[n + c];
        x2 = x[2 * n + c];
    }
    __device__ __forceinline__ bool has_dot() const { return true; }
    __device__ __forceinline__ double row_dot(size_t r3, double y0, double y1, double y2) const
    {
        double p0, p1, p2;
        load(r3, p0, p1, p2);
        return p0 * y0 + p1 * y1 + p2 * y2;
    }
    __device__ __forceinline__ void row_pre(size_t r3, double& p0, double& p1, double& p2) const { load(r3, p0, p1, p2); }
    __device__ __forceinline__ double row_dot_pre(size_t, double p0, double p1, double p2, double y0, double y1, double y2) const { return p0 * y0 + p1 * y1 + p2 * y2; }
};
__device__ __forceinline__ double row_dot_nostore(const XSoA& X, size_t r3, double y0, double y1, double y2) { return X.row_dot(r3, y0, y1, y2); }
__device__ __forceinline__ double row_dot_nostore(const XDir& X, size_t r3, double y0, double y1, double y2) { return X.row_dot_nostore(r3, y0, y1, y2); }

// v_mov_b32_dpp on both halves; BOUND: lanes without a source receive 0, otherwise (and in rows the mask disables) 0 as well (old = 0)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, BOUND);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, BOUND);
    return __hiloint2double(hi, lo);
}
// v of the lane whose byte address (4 * lane) is given (ds_bpermute_b32 on both halves)
__device__ __forceinline__ double lane_gather(double v, int addr)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int V, class XS>
__device__ __forceinline__ void spmv_chunked_static(const int bid, const int nblk, const float* __restrict__ vals, const uint32_t* __restrict__ scol,
                                                    const int32_t* __restrict__ tile_first_row, int64_t n_chunks, const int chunk_tiles, const XS X,
                                                    double* __restrict__ y, double* __restrict__ partials)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double acc = 0.0;
    // XCD-aware placement: consecutive workgroup ids land on different XCDs (round robin over the 8 dies, each with its own L2); give
    // every XCD one contiguous eighth of the chunks so that the x entries its rows gather are shared through that die's L2
    const int pbid = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
    const int64_t n_waves = (int64_t)nblk * 4;
    // The column words (and the first row) of the NEXT tile this wavefront will process are requested behind the current tile's loads: a tile's
    // gather of x then does not wait for a load issued in the same iteration (one memory latency per tile instead of two dependent ones).
    // V & 4: the matrix values come with the non-temporal hint (streamed once per launch: they should not displace x in the L2) — pays when
    // the matrix streams from HBM, costs while it fits the Infinity Cache: chosen by the matrix' size (launch_spmv). Same arithmetic, same bits.
    constexpr bool NT = (V & 4) != 0;
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const int64_t ch0 = (int64_t)pbid * 4 + wave;
    const int64_t n_tiles_chunked = n_chunks * chunk_tiles;
    uint32_t w_next = 0;
    int32_t tfr_next = 0;
    if (ch0 < n_chunks) {
        w_next = scol[ch0 * chunk_tiles * 64 + lane];
        tfr_next = tile_first_row[ch0 * chunk_tiles];
    }
    for (int64_t ch = ch0; ch < n_chunks; ch += n_waves) {
        const int64_t t_begin = ch * chunk_tiles;
        double k0 = 0.0, k1 = 0.0, k2 = 0.0;  // carry into the first segment of the next tile (wave-uniform)
        for (int u = 0; u < chunk_tiles; u++) {
            const int64_t t = t_begin + u;
            const uint32_t w = w_next;
            const int32_t tfr_w = tfr_next;  // bit 31: the tile starts inside a row begun in the previous tile
            const size_t c3 = 3 * (size_t)(w & 0x7fffffffu);
            double x0, x1, x2;
            X.load(c3, x0, x1, x2);
            const float4* q = reinterpret_cast<const float4*>(vals + (size_t)t * 576);
            float4 a, b;
            float cc;
            if (V == 3) {
                a = b = make_float4(1.f, 2.f, 3.f, 4.f);
                cc = 1.f;
            } else if (NT) {
                const nt_f4 ta = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(q + lane)), tb = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(q + 64 + lane));
                a = make_float4(ta.x, ta.y, ta.z, ta.w);
                b = make_float4(tb.x, tb.y, tb.z, tb.w);
                cc = __builtin_nontemporal_load(vals + (size_t)t * 576 + 512 + lane);
            } else {
                a = q[lane];
                b = q[64 + lane];
                cc = vals[(size_t)t * 576 + 512 + lane];
            }
            {
                const int64_t tn = (u + 1 == chunk_tiles) ? (ch + n_waves) * chunk_tiles : t + 1;
                if (tn < n_tiles_chunked) {
                    w_next = scol[tn * 64 + lane];
                    tfr_next = tile_first_row[tn];
                }
            }
            // which lanes end a row, and which row: known from the column words alone, so the row's entries of the dot-product vector are
            // requested NOW, with the gathers (issued after the row sums they were a dependent load at the tail of every tile: 30 us of a
            // 217 us launch on the 8 M-tet matrix, where they come from HBM)
            const bool tail = (w >> 31) != 0;
            const int tfr = tfr_w & 0x7fffffff;
            const bool tile_cont = tfr_w < 0;
            const unsigned long long tails = __ballot(tail);
            const unsigned long long heads = (tails << 1) | 1ull;
            const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            const int start = 63 - __clzll(heads & le);
            const int row = tfr + __popcll(tails & ((1ull << lane) - 1ull));
            double pr0 = 0.0, pr1 = 0.0, pr2 = 0.0;
            if (V != 1 && V != 3 && tail && X.has_dot()) X.row_pre(3 * (size_t)row, pr0, pr1, pr2);
            double y0 = (double)a.x * x0 + (double)a.y * x1 + (double)a.z * x2;
            double y1 = (double)a.w * x0 + (double)b.x * x1 + (double)b.y * x2;
            double y2 = (double)b.z * x0 + (double)b.w * x1 + (double)cc * x2;
            if (V == 1 || V == 3) {  // ablation: loads + block products only
                acc += y0 + y1 + y2;
                continue;
            }
            // Row sums as differences of prefix sums: a plain (unsegmented) inclusive scan over the 64 lanes -- four DPP row shifts inside the
            // 16-lane rows, then row_bcast:15 and row_bcast:31, no conditionals -- and one cross-lane read of the exclusive prefix at the
            // first lane of the lane's row. (The segmented scan this replaces spent two thirds of the loop's 233 instructions on masks,
            // selects and lane reads; the sums of at most 64 blocks differ from the segment sums by rounding errors ~1e-16 of the tile's
            // total, far below the float matrix entries.)
            const double v0 = y0, v1 = y1, v2 = y2;
#define MS_SCAN_STEP(CTRL, RM, BOUND)               \
            {                                       \
                const double u0 = dpp_mov<CTRL, RM, BOUND>(y0), u1 = dpp_mov<CTRL, RM, BOUND>(y1), u2 = dpp_mov<CTRL, RM, BOUND>(y2); \
                y0 += u0; y1 += u1; y2 += u2;       \
            }
            // (Measured and not kept, round 5: the same six steps with the data moved by ds_bpermute instead of v_mov_b32_dpp — the LDS crossbar is
            // idle in this kernel and rocprofv3 shows the VALU 60 % busy with 55 % of the wave cycles in SQ_WAIT_INST_ANY — give identical bits and
            // no gain for the two cross-row steps (165.5 against 166 us at 8 M tets on the same box), a loss for all six (182 us; 23.7 against
            // 20.7 us at 1 M): the crossbar's latency, six dependent round trips per tile, costs more than the issue slots it frees.)
            MS_SCAN_STEP(0x111, 0xf, true)
            MS_SCAN_STEP(0x112, 0xf, true)
            MS_SCAN_STEP(0x114, 0xf, true)
            MS_SCAN_STEP(0x118, 0xf, true)
            MS_SCAN_STEP(0x142, 0xa, false)  // row_bcast:15 -> rows 1 and 3
            MS_SCAN_STEP(0x143, 0xc, false)  // row_bcast:31 -> rows 2 and 3
#undef MS_SCAN_STEP
            {
                // exclusive prefix at the first lane of this lane's row = everything before the row
                const int addr = start << 2;
                const double e0 = y0 - v0, e1 = y1 - v1, e2 = y2 - v2;
                y0 -= lane_gather(e0, addr);
                y1 -= lane_gather(e1, addr);
                y2 -= lane_gather(e2, addr);
            }
            // first segment: take over the carry of the previous tile of this chunk
            if (tile_cont && start == 0) { y0 += k0; y1 += k1; y2 += k2; }
            // last segment open (the row ends in the next tile of the chunk; never at the end of a chunk): hand it on in registers
            if (((tails >> 63) & 1ull) == 0ull) { k0 = read_lane(y0, 63); k1 = read_lane(y1, 63); k2 = read_lane(y2, 63); }
            if (tail) {
                double* yr = y + 3 * (size_t)row;
                yr[0] = y0;
                yr[1] = y1;
                yr[2] = y2;
                if (X.has_dot()) acc += X.row_dot_pre(3 * (size_t)row, pr0, pr1, pr2, y0, y1, y2);
            }
        }
    }
    if (partials) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) partials[bid] = acc;
    }
}
// Rows longer than a chunk (a rigid body attached to very many points): stored after the chunks, one wavefront per row.
template <class XS>
__device__ __forceinline__ void spmv_long_rows(const int bid, const int nblk, const float* __restrict__ vals, const uint32_t* __restrict__ scol, const uint32_t* __restrict__ list,
                                               int n_list, const int64_t* __restrict__ row_ptr, const uint64_t* __restrict__ row_pos, const XS X, double* __restrict__ y,
                                               double* __restrict__ partials)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = bid * 4 + (threadIdx.x >> 6); k < n_list; k += nblk * 4) {
        const int64_t r = list[k];
        const int64_t len = row_ptr[r + 1] - row_ptr[r];
        const size_t base = (size_t)row_pos[r];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int64_t s = lane; s < len; s += 64) {
            const size_t pos = base + (size_t)s;
            const size_t col = scol[pos] & 0x7fffffffu;
            const float* tv = vals + (pos >> 6) * 576;
            const int l = (int)(pos & 63);
            const float4 qa = reinterpret_cast<const float4*>(tv)[l];
            const float4 qb = reinterpret_cast<const float4*>(tv)[64 + l];
            const float cc = tv[512 + l];
            double x0, x1, x2;
            X.load(3 * col, x0, x1, x2);
            a0 += (double)qa.x * x0 + (double)qa.y * x1 + (double)qa.z * x2;
            a1 += (double)qa.w * x0 + (double)qb.x * x1 + (double)qb.y * x2;
            a2 += (double)qb.z * x0 + (double)qb.w * x1 + (double)cc * x2;
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if (lane == 0) {
            double* yr = y + 3 * (size_t)r;
            yr[0] = a0;
            yr[1] = a1;
            yr[2] = a2;
            if (X.has_dot()) acc += X.row_dot(3 * (size_t)r, a0, a1, a2);
        }
    }
    if (partials) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) partials[bid] = acc;
    }
}

static int spmv_grid(const Context& c, int64_t n_chunks, int max_grid)
{
    // one wavefront per chunk when they fit the grid cap; a multiple of 8 workgroups keeps the XCD placement of spmv_chunked_static
    const int cap = std::min(c.spmv_grid_cap > 0 ? c.spmv_grid_cap : 2048, max_grid);
    return (int)std::max<int64_t>(std::min<int64_t>(((n_chunks + 3) / 4 + 7) / 8 * 8, cap / 8 * 8), 8);
}
// SpMV of the contact part: row sums of A_dyn x. Its block rows are short (a contact touches a handful of nodes: four lanes per row)
// except the rows of rigid bodies in contact, which hold one block per touching node (thousands): those are cut into chunks of
// <= CHUNK_BLOCKS blocks, one wavefront per chunk. Row sums go to `yd` (one per compact row), the chunks of a multi-chunk row to
// `chunk_partial`; the consumer (dyn_row) adds them to y in a fixed order (deterministic, no atomics). p . (A_dyn x) is linear in
// the rows and chunks and summed right here.
struct DynPart  // the contact part as the fused SpMV kernel sees it
{
    const float* vals;
    const uint32_t* colw;
    const int64_t* row_ptr;
    const uint32_t* row_chunk0;
    const int32_t* chunk_row;
    const int32_t* rowmap;
    double* yd;             // 3 per compact row (rows with a single chunk)
    double* chunk_partial;  // 3 per chunk (rows with several chunks)
    int64_t n_chunks;
    int64_t n_rows;
};
template <class XS>
__device__ __forceinline__ void spmv_chunks(const int bid, const int nblk, const DynPart& d, const XS X, double* __restrict__ partials)
{
    // These few workgroups run beside thousands of static-part wavefronts that saturate the memory system, where every dependent load
    // costs 1.5-2 us: their chains must be short or they become the critical path of the whole launch (measured: +4.5 us with one lane
    // per row and the chunk loop behind it). Workgroups [0, g_chunks) reduce the chunks of long rows, one per wavefront; the others take
    // the short rows, four lanes per row.
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63;
    const int g_chunks = (int)min((d.n_chunks + 3) / 4, (int64_t)nblk / 2);
    double dot = 0.0;
    if (bid < g_chunks) {
        for (int64_t ch = (int64_t)bid * 4 + (threadIdx.x >> 6); ch < d.n_chunks; ch += (int64_t)g_chunks * 4) {
            const int r = d.chunk_row[ch];
            const uint32_t c0 = d.row_chunk0[r], c1 = d.row_chunk0[r + 1];
            const int64_t s0 = d.row_ptr[r] + (int64_t)(ch - c0) * CHUNK_BLOCKS;
            const int64_t s1 = min(d.row_ptr[r + 1], s0 + (int64_t)CHUNK_BLOCKS);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int64_t s = s0 + lane; s < s1; s += 64) {
                const size_t col = (size_t)(d.colw[s] & 0x7fffffffu);
                const float* tv = d.vals + (size_t)(s >> 6) * 576;
                const int l = (int)(s & 63);
                const float4 qa = reinterpret_cast<const float4*>(tv)[l];
                const float4 qb = reinterpret_cast<const float4*>(tv)[64 + l];
                const float cc = tv[512 + l];
                double x0, x1, x2;
                X.load(3 * col, x0, x1, x2);
                a0 += (double)qa.x * x0 + (double)qa.y * x1 + (double)qa.z * x2;
                a1 += (double)qa.w * x0 + (double)qb.x * x1 + (double)qb.y * x2;
                a2 += (double)qb.z * x0 + (double)qb.w * x1 + (double)cc * x2;
            }
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0) {
                double* out = (c1 - c0 == 1) ? d.yd + 3 * (size_t)r : d.chunk_partial + 3 * (size_t)ch;
                out[0] = a0;
                out[1] = a1;
                out[2] = a2;
                if (X.has_dot()) dot += row_dot_nostore(X, 3 * (size_t)d.rowmap[r], a0, a1, a2);
            }
        }
    } else {
        const int g_short = nblk - g_chunks;
        const int q = threadIdx.x & 3;
        for (int64_t r = (int64_t)(bid - g_chunks) * (BLOCK / 4) + (threadIdx.x >> 2); r < d.n_rows; r += (int64_t)g_short * (BLOCK / 4)) {
            const int64_t s0 = d.row_ptr[r], s1 = d.row_ptr[r + 1];
            const bool is_short = s1 - s0 <= DYN_SHORT_ROW;  // (the same for the four lanes of a row)
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            if (is_short) {
                for (int64_t s = s0 + q; s < s1; s += 4) {
                    const size_t col = (size_t)(d.colw[s] & 0x7fffffffu);
                    const float* tv = d.vals + (size_t)(s >> 6) * 576;
                    const int l = (int)(s & 63);
                    const float4 qa = reinterpret_cast<const float4*>(tv)[l];
                    const float4 qb = reinterpret_cast<const float4*>(tv)[64 + l];
                    const float cc = tv[512 + l];
                    double x0, x1, x2;
                    X.load(3 * col, x0, x1, x2);
                    a0 += (double)qa.x * x0 + (double)qa.y * x1 + (double)qa.z * x2;
                    a1 += (double)qa.w * x0 + (double)qb.x * x1 + (double)qb.y * x2;
                    a2 += (double)qb.z * x0 + (double)qb.w * x1 + (double)cc * x2;
                }
            }
            // the four partial sums of a row, in a fixed order (all lanes of the wavefront take part in the shuffles)
            a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
            a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
            if (is_short && q == 0) {
                double* out = d.yd + 3 * (size_t)r;
                out[0] = a0;
                out[1] = a1;
                out[2] = a2;
                if (X.has_dot()) dot += row_dot_nostore(X, 3 * (size_t)d.rowmap[r], a0, a1, a2);
            }
        }
    }
    if (partials) {
        dot = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[bid] = dot;
    }
}
// contribution of the contact part to block row `row` (written by spmv_chunks): short rows (no chunk) and single-chunk rows are the row sum itself,
// longer rows the sum of their chunk partials in ascending order.
// Rows may have many chunk partials (a rigid body under 10^5 contacts: ~270 chunks): CALLED BY ALL LANES OF A WAVEFRONT (lanes
// without a row pass row = -1). Rows up to DYN_FOLD_SERIAL chunks are folded by their own lane; a longer row is folded by the
// whole wavefront — lane l adds chunks l, l + 64, ... in ascending order, then the fixed-shape wave_sum: deterministic, the same bits in every
// kernel that consumes the contact part (one lane walking 270 chunks held k_pcg_step at 30 us on configs[2]; the SpMV beside it takes 8).
constexpr uint32_t DYN_FOLD_SERIAL = 8;
__device__ __forceinline__ void dyn_row_wave(const int32_t* __restrict__ crow_of_row, const uint32_t* __restrict__ row_chunk0, const double* __restrict__ yd,
                                             const double* __restrict__ chunk_partial, int64_t row, double& q0, double& q1, double& q2)
{
    uint32_t c0 = 0, c1 = 0;
    int32_t cr = -1;
    if (row >= 0) {
        cr = crow_of_row[row];
        if (cr >= 0) {
            c0 = row_chunk0[cr];
            c1 = row_chunk0[cr + 1];
        }
    }
    const bool lng = c1 - c0 > DYN_FOLD_SERIAL;
    if (cr >= 0 && !lng) {
        if (c1 - c0 <= 1) {
            q0 += yd[3 * (size_t)cr];
            q1 += yd[3 * (size_t)cr + 1];
            q2 += yd[3 * (size_t)cr + 2];
        } else {
            for (uint32_t k = c0; k < c1; k++) {
                q0 += chunk_partial[3 * (size_t)k];
                q1 += chunk_partial[3 * (size_t)k + 1];
                q2 += chunk_partial[3 * (size_t)k + 2];
            }
        }
    }
    unsigned long long mask = __ballot(lng);
    const int lane = threadIdx.x & 63;
    while (mask) {  // (wave-uniform)
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const uint32_t b0 = (uint32_t)__shfl((int)c0, src, 64), b1 = (uint32_t)__shfl((int)c1, src, 64);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (uint32_t k = b0 + (uint32_t)lane; k < b1; k += 64) {
            a0 += chunk_partial[3 * (size_t)k];
            a1 += chunk_partial[3 * (size_t)k + 1];
            a2 += chunk_partial[3 * (size_t)k + 2];
        }
        a0 = read_lane(wave_sum(a0), 0);
        a1 = read_lane(wave_sum(a1), 0);
        a2 = read_lane(wave_sum(a2), 0);
        if (lane == src) {
            q0 += a0;
            q1 += a1;
            q2 += a2;
        }
    }
}
struct StaticPart  // the static part as the fused SpMV kernel sees it
{
    const float* vals;
    const uint32_t* scol;
    const int32_t* tile_first_row;
    const uint32_t* long_rows;
    const int64_t* row_ptr;
    const uint64_t* row_pos;
    int64_t n_chunks;
    int n_long_rows;
    int chunk_tiles;
};
// One launch for y = A_static x (rows written once, see spmv_chunked_static) and the contact part's row sums (yd / chunk_partial); the
// consumer adds them (k_pcg_step inside the solver, k_spmv_combine elsewhere). The few workgroups of the contact part and of over-long
// rows come FIRST in the grid: dispatched last they would start when the static part drains and add their whole duration to the kernel
// (measured: 27.7 us with them at the end, 21.7 us for the static part alone). Workgroups [0, g1): chunks of the contact part,
// [g1, g1 + gr): over-long static rows, the rest: chunks of the static part. partials keep the order static | long | contact.
template <int V>
__global__ __launch_bounds__(BLOCK) void k_spmv_fused(int g0, int gr, int g1, StaticPart m, DynPart d, const double* __restrict__ x, double* __restrict__ y,
                                                     const double* __restrict__ pdot, double* __restrict__ partials, const PcgCtrl* __restrict__ ctrl,
                                                     uint64_t* __restrict__ clk)
{
    if (ctrl && ctrl->done) return;
    const int b = (int)blockIdx.x;
    // sampled launches (clk != null, pinned host memory): every workgroup records when it started and finished on the device's constant
    // clock; the host takes max(end) - min(start), the launch's execution time without anything the stream does around it
    const uint64_t t_start = clk ? wall_clock64() : 0;
    const XPlain X{x, pdot};
    if (b < g1) spmv_chunks(b, g1, d, X, partials ? partials + g0 + gr : nullptr);
    else if (b < g1 + gr) spmv_long_rows(b - g1, gr, m.vals, m.scol, m.long_rows, m.n_long_rows, m.row_ptr, m.row_pos, X, y, partials ? partials + g0 : nullptr);
    else spmv_chunked_static<V>(b - g1 - gr, g0, m.vals, m.scol, m.tile_first_row, m.n_chunks, m.chunk_tiles, X, y, partials);
    if (clk) {
        __syncthreads();
        if (threadIdx.x == 0) {
            clk[2 * b] = t_start;
            clk[2 * b + 1] = wall_clock64();
        }
    }
}
#ifdef MISTARK_BENCH_VARIANTS  // measurement-only kernels (tools/spmv_sweep.py): make BENCH_VARIANTS=1
// measurement only (spmv_variant 12): the static part with SoA input (see XSoA); y stays interleaved
__global__ __launch_bounds__(BLOCK) void k_spmv_soa(int g0, StaticPart m, XSoA X, double* __restrict__ y, double* __restrict__ partials)
{
    spmv_chunked_static<0>((int)blockIdx.x, g0, m.vals, m.scol, m.tile_first_row, m.n_chunks, m.chunk_tiles, X, y, partials);
}
__global__ __launch_bounds__(BLOCK) void k_to_soa(const double* __restrict__ v, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    out[i] = v[3 * i];
    out[n + i] = v[3 * i + 1];
    out[2 * n + i] = v[3 * i + 2];
}
#endif
// The PCG's iteration k as the solver launches it: what k_pcg_dir did for iteration k-1 (sums of r.r and r.z, convergence test, beta) in the
// prologue of every workgroup (all of them compute the same numbers from the same partial sums; workgroup 0 records them), then
// q = A p with p = z + beta p_old formed on the fly and stored by the lanes that finish a row.
struct DirArgs
{
    const double* z;
    const double* pold;
    double* pnew;
    const double* part_rr;
    const double* part_rz;
    int nparts, k;
    double abs_tol, rel_tol;
};
// convergence test and beta from the partial sums step k-1 left; returns false when the solve is over (and records why)
__device__ __forceinline__ bool pcg_direction(const DirArgs& a, PcgCtrl* __restrict__ ctrl, double* sm, double& beta)
{
    const int done = ctrl->done;
    if (done == 1) return false;
    if (done == 2) {  // indefiniteness stop decided in k_pcg_step of the previous iteration
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->done = 1;
        return false;
    }
    beta = 0.0;
    if (a.k == 1) return true;  // p_1 = z_0
    const int kp = a.k - 1;     // the iteration whose step left the partial sums
    const double rr = sum_partials(a.part_rr, a.nparts, sm);
    const double rz_new = sum_partials(a.part_rz, a.nparts, sm);
    const double error = sqrt(rr / ctrl->bb);
    const bool conv = error < a.abs_tol || error / 1.0 < a.rel_tol;  // error_0 = 1 for x0 = 0
    if (conv) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctrl->error = error;
            ctrl->n_iter = kp;
            ctrl->converged = 1;
            ctrl->done = 1;
        }
        return false;
    }
    beta = rz_new / ctrl->rz[kp & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->rz[(kp + 1) & 1] = rz_new;
        ctrl->error = error;
        ctrl->n_iter = kp;
    }
    return true;
}
__global__ __launch_bounds__(BLOCK) void k_spmv_dir(int g0, int gr, int g1, StaticPart m, DynPart d, DirArgs a, double* __restrict__ y, double* __restrict__ partials,
                                                   PcgCtrl* __restrict__ ctrl)
{
    __shared__ double sm[4];
    double beta;
    if (!pcg_direction(a, ctrl, sm, beta)) return;
    const int b = (int)blockIdx.x;
    const XDir X{a.z, a.pold, a.pnew, beta};
    if (b < g1) spmv_chunks(b, g1, d, X, partials + g0 + gr);
    else if (b < g1 + gr) spmv_long_rows(b - g1, gr, m.vals, m.scol, m.long_rows, m.n_long_rows, m.row_ptr, m.row_pos, X, y, partials + g0);
    else spmv_chunked_static<0>(b - g1 - gr, g0, m.vals, m.scol, m.tile_first_row, m.n_chunks, m.chunk_tiles, X, y, partials);
}
// the same test at the end of a batch of iterations (the host looks at the control block there)
__global__ __launch_bounds__(BLOCK) void k_pcg_check(DirArgs a, PcgCtrl* __restrict__ ctrl)
{
    __shared__ double sm[4];
    double beta;
    (void)pcg_direction(a, ctrl, sm, beta);
}
__global__ __launch_bounds__(BLOCK) void k_spmv_combine(int64_t nbr, const int32_t* __restrict__ crow_of_row, const uint32_t* __restrict__ row_chunk0,
                                                       const double* __restrict__ yd, const double* __restrict__ chunk_partial, double* __restrict__ y)
{
    const int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    dyn_row_wave(crow_of_row, row_chunk0, yd, chunk_partial, row < nbr ? row : -1, q0, q1, q2);
    if (row >= nbr) return;
    y[3 * row] += q0;
    y[3 * row + 1] += q1;
    y[3 * row + 2] += q2;
}
// y = (A_static + A_dynamic) x; partial sums of pdot . y go to partials[0 .. return value)
// grid of the fused SpMV launch on the current matrix: workgroups of the static chunks, of over-long static rows, of the contact part
static void spmv_launch_shape(Context& c, int& g0, int& gr, int& g1, StaticPart& sp, DynPart& d)
{
    const BsrPart& m0 = c.part[0];
    BsrPart& m1 = c.part[1];
    g0 = spmv_grid(c, m0.n_chunks_static, MAX_PARTIALS / 2);
    gr = std::min(((m0.n_long_rows + 3) / 4 + 7) / 8 * 8, MAX_PARTIALS / 4);
    sp = StaticPart{m0.vals.p, m0.scol.p, m0.tile_first_row.p, m0.long_rows.p, m0.row_ptr.p, m0.row_pos.p, m0.n_chunks_static, m0.n_long_rows, m0.chunk_tiles};
    d = DynPart{};
    g1 = 0;
    if (m1.nnzb > 0) {
        // workgroups for the chunks of long rows (one per wavefront) + for the short rows (four lanes each); a multiple of 8 keeps the XCD placement of the static part
        g1 = (int)std::min<int64_t>(((m1.n_chunks + 3) / 4 + (m1.n_rows + BLOCK / 4 - 1) / (BLOCK / 4) + 7) / 8 * 8, MAX_PARTIALS / 4);
        d = DynPart{m1.vals.p, m1.colw.p, m1.row_ptr.p, m1.row_chunk0.p, m1.chunk_row.p, m1.rowmap.p, m1.yd.p, m1.chunk_partial.p, m1.n_chunks, m1.n_rows};
    }
}
template <int V>
static int launch_spmv(Context& c, const double* x, double* y, const double* pdot, double* partials, const PcgCtrl* ctrl, bool combine = true, uint64_t* clk = nullptr)
{
    BsrPart& m1 = c.part[1];
    int g0, gr, g1;
    StaticPart sp;
    DynPart d;
    spmv_launch_shape(c, g0, gr, g1, sp, d);
    if (c.spmv_variant == 30) g1 = gr = 0;  // (measurement: the static chunks alone through the same kernel)
    // non-temporal value loads once the matrix cannot stay in the 256 MiB Infinity Cache beside the vectors (option spmv_nt: -1 = by size, 0 / 1)
    const bool nt = V == 0 && (c.spmv_nt >= 0 ? c.spmv_nt != 0 : (size_t)c.part[0].ntiles * 2304 + (size_t)c.part[0].ntiles * 256 > ((size_t)160 << 20));
if (nt) hipLaunchKernelGGL(k_spmv_fused<4>, dim3(g0 + gr + g1), dim3(BLOCK), 0, c.stream, g0, gr, g1, sp, d, x, y, pdot, partials, ctrl, clk);
    else hipLaunchKernelGGL(k_spmv_fused<V>, dim3(g0 + gr + g1), dim3(BLOCK), 0, c.stream, g0, gr, g1, sp, d, x, y, pdot, partials, ctrl, clk);
    if (g1 > 0 && combine)
        hipLaunchKernelGGL(k_spmv_combine, dim3(grid_for(c.mrows())), dim3(BLOCK), 0, c.stream, c.mrows(), (const int32_t*)m1.crow_of_row.p, (const uint32_t*)m1.row_chunk0.p,
                           (const double*)m1.yd.p, (const double*)m1.chunk_partial.p, y);
    return g0 + gr + g1;
}
static int launch_spmv_dir(Context& c, const DirArgs& a, double* y, double* partials)
{
    const BsrPart& m0 = c.part[0];
    BsrPart& m1 = c.part[1];
    const int g0 = spmv_grid(c, m0.n_chunks_static, MAX_PARTIALS / 2);
    const int gr = std::min(((m0.n_long_rows + 3) / 4 + 7) / 8 * 8, MAX_PARTIALS / 4);
    const StaticPart sp{m0.vals.p, m0.scol.p, m0.tile_first_row.p, m0.long_rows.p, m0.row_ptr.p, m0.row_pos.p, m0.n_chunks_static, m0.n_long_rows, m0.chunk_tiles};
    DynPart d{};
    int g1 = 0;
    if (m1.nnzb > 0) {
        g1 = (int)std::min<int64_t>(((m1.n_chunks + 3) / 4 + (m1.n_rows + BLOCK / 4 - 1) / (BLOCK / 4) + 7) / 8 * 8, MAX_PARTIALS / 4);
        d = DynPart{m1.vals.p, m1.colw.p, m1.row_ptr.p, m1.row_chunk0.p, m1.chunk_row.p, m1.rowmap.p, m1.yd.p, m1.chunk_partial.p, m1.n_chunks, m1.n_rows};
    }
    hipLaunchKernelGGL(k_spmv_dir, dim3(g0 + gr + g1), dim3(BLOCK), 0, c.stream, g0, gr, g1, sp, d, a, y, partials, c.ctrl.p);
    return g0 + gr + g1;
}
#ifdef MISTARK_BENCH_VARIANTS
// reference point for the micro-benchmark (variant 9): a plain grid-stride float4 read of the matrix values, i.e. what streaming the
// matrix costs at best on this box (measured 16.2 us for the 1M-tet block = 6.3 TB/s)
__global__ __launch_bounds__(BLOCK) void k_stream_ref(const float4* __restrict__ v, size_t n4, double* __restrict__ partials)
{
    __shared__ double sm[4];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLOCK) {
        const float4 a = v[i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const double t = block_sum((double)(s.x + s.y + s.z + s.w), sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// variant 11: what a block product needs (column word, values, x gather, nine FMAs) in the simplest possible loop, no row reduction:
// the floor for any kernel on this storage
__global__ __launch_bounds__(BLOCK) void k_spmv_products_only(const float* __restrict__ vals, const uint32_t* __restrict__ colw, int64_t ntiles, const double* __restrict__ x,
                                                             double* __restrict__ partials)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t tpw = (ntiles + n_waves - 1) / n_waves, t0 = gw * tpw, t1 = t0 + tpw < ntiles ? t0 + tpw : ntiles;
    double acc = 0.0;
    for (int64_t t = t0; t < t1; t++) {
        const uint32_t w = colw[t * 64 + lane];
        const float4* q = reinterpret_cast<const float4*>(vals + (size_t)t * 576);
        const float4 a = q[lane], b = q[64 + lane];
        const float cc = vals[(size_t)t * 576 + 512 + lane];
        const size_t c3 = 3 * (size_t)(w & 0x7fffffffu);
        const double x0 = x[c3], x1 = x[c3 + 1], x2 = x[c3 + 2];
        acc += ((double)a.x * x0 + (double)a.y * x1 + (double)a.z * x2) + ((double)a.w * x0 + (double)b.x * x1 + (double)b.y * x2) +
               ((double)b.z * x0 + (double)b.w * x1 + (double)cc * x2);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
#endif
// Micro-benchmark of the SpMV kernel on the assembled matrix: n back-to-back launches of q = A p (+ fused dot), HIP events
// around the whole batch on the engine's stream. Returns the average launch duration in microseconds.
double spmv_bench(Context& c, int n)
{
    if (!c.have_matrix) throw Error("spmv_bench: matrix not assembled");
    hipEvent_t e0, e1;
    MS_CHECK(hipEventCreate(&e0));
    MS_CHECK(hipEventCreate(&e1));
    vec_fill(c, c.p.p, 1.0, c.ndofs);
    for (int w = 0; w < 3; w++) launch_spmv<0>(c, c.p.p, c.q.p, c.p.p, c.partials.p, nullptr);
    MS_CHECK(hipEventRecord(e0, c.stream));
    for (int i = 0; i < n; i++) {
        switch (c.spmv_variant) {
#ifdef MISTARK_BENCH_VARIANTS
            case 1: launch_spmv<1>(c, c.p.p, c.q.p, c.p.p, c.partials.p, nullptr, false); break;
            case 11: hipLaunchKernelGGL(k_spmv_products_only, dim3(c.spmv_grid_cap > 0 ? c.spmv_grid_cap : 1024), dim3(BLOCK), 0, c.stream, (const float*)c.part[0].vals.p, (const uint32_t*)c.part[0].scol.p, c.part[0].ntiles, (const double*)c.p.p, c.partials.p); break;
            case 9: hipLaunchKernelGGL(k_stream_ref, dim3(2048), dim3(BLOCK), 0, c.stream, (const float4*)c.part[0].vals.p, (size_t)c.part[0].ntiles * 144, c.partials.p); break;
            case 3: launch_spmv<3>(c, c.p.p, c.q.p, c.p.p, c.partials.p, nullptr, false); break;
            case 12: {  // SoA input vector (static part only; compare with variant 0 on a contact-free matrix or read it as a lower bound)
                int g0, gr, g1;
                StaticPart sp;
                DynPart d;
                spmv_launch_shape(c, g0, gr, g1, sp, d);
                if (i == 0) hipLaunchKernelGGL(k_to_soa, dim3(grid_for(c.nbr)), dim3(BLOCK), 0, c.stream, (const double*)c.p.p, c.nbr, c.tmp_a.p);
                hipLaunchKernelGGL(k_spmv_soa, dim3(g0), dim3(BLOCK), 0, c.stream, g0, sp, XSoA{c.tmp_a.p, (size_t)c.nbr}, c.q.p, c.partials.p);
                break;
            }
#else
            case 1: case 3: case 9: case 11: case 12: throw Error("spmv_bench: the measurement-only variants are not in this build (make -C stark_amd/csrc BENCH_VARIANTS=1)");
#endif
            default: launch_spmv<0>(c, c.p.p, c.q.p, c.p.p, c.partials.p, nullptr, false);  // as inside the solver: k_pcg_step adds the contact rows
        }
    }
    MS_CHECK(hipEventRecord(e1, c.stream));
    MS_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MS_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 1000.0 * ms / n;
}

void spmv_device(Context& c, const double* x, double* y, const double* pdot, double* partials, bool timed)
{
    (void)timed;
    launch_spmv<0>(c, x, y, pdot, partials, nullptr);
}

// ======================================================================================================================
// PCG (BlockedSparseMatrix/solve_pcg.h:83-232), x0 = 0. Iteration k = 1..max_iter is three launches:
//   k_spmv      q = A p, partial p.q
//   k_pcg_step  alpha = rz/pAp; x += alpha p; r -= alpha q; z = M^-1 r; partial r.r, r.z      (indefiniteness test)
//   k_pcg_dir   error = sqrt(rr/bb); convergence test; beta = rz'/rz; p = z + beta p
// Scalars never leave the device inside the loop; `ctrl->done` turns the remaining launches of a batch into no-ops.
// ======================================================================================================================
__global__ __launch_bounds__(BLOCK) void k_pcg_init(const double* __restrict__ b, const float* __restrict__ dinv, int64_t nbr, double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ z, double* __restrict__ p, double* __restrict__ part_bb, double* __restrict__ part_rz)
{
    __shared__ double sm[4];
    double bb = 0.0, rz = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < nbr; row += (int64_t)gridDim.x * BLOCK) {
        const double r0 = b[3 * row], r1 = b[3 * row + 1], r2 = b[3 * row + 2];
        const float* d = dinv + 9 * row;
        // column-major-agnostic: the inverse is symmetric
        const double z0 = (double)d[0] * r0 + (double)d[1] * r1 + (double)d[2] * r2;
        const double z1 = (double)d[3] * r0 + (double)d[4] * r1 + (double)d[5] * r2;
        const double z2 = (double)d[6] * r0 + (double)d[7] * r1 + (double)d[8] * r2;
        x[3 * row] = 0.0; x[3 * row + 1] = 0.0; x[3 * row + 2] = 0.0;
        r[3 * row] = r0; r[3 * row + 1] = r1; r[3 * row + 2] = r2;
        z[3 * row] = z0; z[3 * row + 1] = z1; z[3 * row + 2] = z2;
        p[3 * row] = z0; p[3 * row + 1] = z1; p[3 * row + 2] = z2;
        bb += r0 * r0 + r1 * r1 + r2 * r2;
        rz += r0 * z0 + r1 * z1 + r2 * z2;
    }
    bb = block_sum(bb, sm);
    rz = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        part_bb[blockIdx.x] = bb;
        part_rz[blockIdx.x] = rz;
    }
}
// The prologue of a solve in one launch plus k_pcg_init2 (single-GPU path): b = scale * rhs (the Newton loop solves A du = -g), the
// block-Jacobi preconditioner of the rows (k_block_diag_inverse), x = 0, r = b, z = p = M^-1 r. Before: negation, preconditioner and
// k_pcg_init as three launches with their boundaries. (Folding k_pcg_init2 in as well — the workgroup that draws the last ticket adds
// the partial sums — was measured and is slower: 674 same-address atomics serialise at ~50 ns each.)
// (src_row: the right-hand side is in the caller's numbering, the solve in the solver's: Context::perm_active)
__global__ __launch_bounds__(BLOCK) void k_pcg_prologue(const double* __restrict__ rhs, double scale, const float* __restrict__ vals, const int32_t* __restrict__ diag_slot,
                                                        const float* __restrict__ vals_dyn, const int32_t* __restrict__ diag_slot_dyn, int64_t nbr, float* __restrict__ dinv,
                                                        double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, double* __restrict__ p, double* __restrict__ part_bb,
                                                        double* __restrict__ part_rz, const int32_t* __restrict__ src_row)
{
    __shared__ double sm[4];
    double bb = 0.0, rz = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < nbr; row += (int64_t)gridDim.x * BLOCK) {
        float m[9], d[9];
        const uint32_t s = (uint32_t)diag_slot[row];
#pragma unroll
        for (int k = 0; k < 9; k++) m[k] = vals[tile_val_index(s, k)];
        if (vals_dyn) {
            const int32_t sd = diag_slot_dyn[row];
            if (sd >= 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) m[k] += vals_dyn[tile_val_index((uint32_t)sd, k)];
            }
        }
        sym3_inverse(m, d);
#pragma unroll
        for (int k = 0; k < 9; k++) dinv[9 * row + k] = d[k];
        const int64_t g = src_row ? (int64_t)src_row[row] : row;
        const double r0 = scale * rhs[3 * g], r1 = scale * rhs[3 * g + 1], r2 = scale * rhs[3 * g + 2];
        const double z0 = (double)d[0] * r0 + (double)d[1] * r1 + (double)d[2] * r2;
        const double z1 = (double)d[3] * r0 + (double)d[4] * r1 + (double)d[5] * r2;
        const double z2 = (double)d[6] * r0 + (double)d[7] * r1 + (double)d[8] * r2;
        x[3 * row] = 0.0; x[3 * row + 1] = 0.0; x[3 * row + 2] = 0.0;
        r[3 * row] = r0; r[3 * row + 1] = r1; r[3 * row + 2] = r2;
        z[3 * row] = z0; z[3 * row + 1] = z1; z[3 * row + 2] = z2;
        p[3 * row] = z0; p[3 * row + 1] = z1; p[3 * row + 2] = z2;
        bb += r0 * r0 + r1 * r1 + r2 * r2;
        rz += r0 * z0 + r1 * z1 + r2 * z2;
    }
    bb = block_sum(bb, sm);
    rz = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        part_bb[blockIdx.x] = bb;
        part_rz[blockIdx.x] = rz;
    }
}
// vector in solver numbering -> the caller's numbering (dst_row = Context::iperm), and back (k_rows_to_solver)
__global__ __launch_bounds__(BLOCK) void k_rows_from_solver(const double* __restrict__ v, const int32_t* __restrict__ dst_row, int64_t nbr, double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= 3 * nbr) return;
    const int64_t row = t / 3;
    out[3 * (int64_t)dst_row[row] + (t - 3 * row)] = v[t];
}
__global__ __launch_bounds__(BLOCK) void k_rows_to_solver(const double* __restrict__ v, const int32_t* __restrict__ src_row, int64_t nbr, double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= 3 * nbr) return;
    const int64_t row = t / 3;
    out[t] = v[3 * (int64_t)src_row[row] + (t - 3 * row)];
}
void rows_from_solver(Context& c, const double* v_solver, double* v_caller)
{
    hipLaunchKernelGGL(k_rows_from_solver, dim3(grid_for(3 * c.nbr)), dim3(BLOCK), 0, c.stream, v_solver, (const int32_t*)c.iperm.p, c.nbr, v_caller);
}
void rows_to_solver(Context& c, const double* v_caller, double* v_solver)
{
    hipLaunchKernelGGL(k_rows_to_solver, dim3(grid_for(3 * c.nbr)), dim3(BLOCK), 0, c.stream, v_caller, (const int32_t*)c.iperm.p, c.nbr, v_solver);
}
__global__ __launch_bounds__(BLOCK) void k_pcg_init2(const double* __restrict__ part_bb, const double* __restrict__ part_rz, int nparts, double abs_tol, PcgCtrl* __restrict__ ctrl,
                                                     int stride)
{
    __shared__ double sm[4];
    const double bb = sum_partials(part_bb, nparts, sm, stride);
    const double rz = sum_partials(part_rz, nparts, sm, stride);
    if (threadIdx.x == 0) {
        ctrl->bb = bb;
        ctrl->rz[1] = rz;
        ctrl->rz[0] = 0.0;
        ctrl->indef = 0;
        ctrl->n_iter = 0;
        ctrl->converged = 0;
        ctrl->done = 0;
        ctrl->error = 1.0;  // r = b  =>  error_0 = 1
        if (bb < abs_tol * abs_tol) {  // zero right-hand side (solve_pcg.h:125-131)
            ctrl->done = 1;
            ctrl->converged = 1;
            ctrl->error = 0.0;
        } else if (1.0 < abs_tol) {    // initial residual already below tolerance (:150-156)
            ctrl->done = 1;
            ctrl->converged = 1;
        }
    }
}
// (The loads of a thread's first block row are issued BEFORE the reduction of the partial sums every workgroup starts with: that
// reduction is a chain of dependent steps of 1.5-2 us during which the memory system would otherwise idle; with one row per thread, which is
// how the solver sizes the grid, that is all of the kernel's loads.)
struct StepRow
{
    double q0, q1, q2, r0, r1, r2, x0, x1, x2, p0, p1, p2;
    float d[9];
};
__device__ __forceinline__ void step_load(StepRow& w, int64_t row, const float* __restrict__ dinv, const double* __restrict__ p, const double* __restrict__ q,
                                          const double* __restrict__ x, const double* __restrict__ r, const int32_t* __restrict__ crow_of_row,
                                          const uint32_t* __restrict__ row_chunk0, const double* __restrict__ yd, const double* __restrict__ chunk_partial)
{
    const size_t i = 3 * (size_t)row;
    w.q0 = q[i]; w.q1 = q[i + 1]; w.q2 = q[i + 2];
    w.r0 = r[i]; w.r1 = r[i + 1]; w.r2 = r[i + 2];
    w.x0 = x[i]; w.x1 = x[i + 1]; w.x2 = x[i + 2];
    w.p0 = p[i]; w.p1 = p[i + 1]; w.p2 = p[i + 2];
#pragma unroll
    for (int u = 0; u < 9; u++) w.d[u] = dinv[9 * row + u];
    // (+ the contact part of q: dyn_row_wave, called by the whole wavefront behind this)
}
__global__ __launch_bounds__(BLOCK) void k_pcg_step(int k, int stop_on_indef, const double* __restrict__ part_pq, int n_pq, const float* __restrict__ dinv, int64_t nbr,
                                                    const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ z, double* __restrict__ part_rr, double* __restrict__ part_rz, PcgCtrl* __restrict__ ctrl,
                                                    const int32_t* __restrict__ crow_of_row, const uint32_t* __restrict__ row_chunk0, const double* __restrict__ yd,
                                                    const double* __restrict__ chunk_partial)
{
    const int done = ctrl->done;
    const double rz = ctrl->rz[k & 1];
    int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    StepRow w;
    if (row < nbr) step_load(w, row, dinv, p, q, x, r, crow_of_row, row_chunk0, yd, chunk_partial);
    if (crow_of_row) dyn_row_wave(crow_of_row, row_chunk0, yd, chunk_partial, row < nbr ? row : -1, w.q0, w.q1, w.q2);  // + contact part (k_spmv_fused)
    if (done) return;
    __shared__ double sm[4];
    const double pAp = sum_partials(part_pq, n_pq, sm);
    if (pAp <= 0.0) {
        if (stop_on_indef) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctrl->indef = 1;
                ctrl->n_iter = k;
                ctrl->converged = 0;
                ctrl->done = 2;  // becomes visible to the next kernel
            }
            // all blocks take the same decision: leave x untouched
            if (threadIdx.x == 0) {
                part_rr[blockIdx.x] = 0.0;
                part_rz[blockIdx.x] = 0.0;
            }
            return;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->indef = 1;
    }
    const double alpha = rz / pAp;
    double rr = 0.0, rzn = 0.0;
    for (; __any(row < nbr);) {  // (wave-uniform: dyn_row_wave needs the whole wavefront)
        if (row < nbr) {
            const size_t i = 3 * (size_t)row;
            const double r0 = w.r0 - alpha * w.q0, r1 = w.r1 - alpha * w.q1, r2 = w.r2 - alpha * w.q2;
            x[i] = w.x0 + alpha * w.p0;
            x[i + 1] = w.x1 + alpha * w.p1;
            x[i + 2] = w.x2 + alpha * w.p2;
            r[i] = r0; r[i + 1] = r1; r[i + 2] = r2;
            const float* d = w.d;
            const double z0 = (double)d[0] * r0 + (double)d[1] * r1 + (double)d[2] * r2;
            const double z1 = (double)d[3] * r0 + (double)d[4] * r1 + (double)d[5] * r2;
            const double z2 = (double)d[6] * r0 + (double)d[7] * r1 + (double)d[8] * r2;
            z[i] = z0; z[i + 1] = z1; z[i + 2] = z2;
            rr += r0 * r0 + r1 * r1 + r2 * r2;
            rzn += r0 * z0 + r1 * z1 + r2 * z2;
            row += (int64_t)gridDim.x * BLOCK;
            if (row < nbr) step_load(w, row, dinv, p, q, x, r, crow_of_row, row_chunk0, yd, chunk_partial);
        }
        if (crow_of_row && __any(row < nbr)) dyn_row_wave(crow_of_row, row_chunk0, yd, chunk_partial, row < nbr ? row : -1, w.q0, w.q1, w.q2);
    }
    rr = block_sum(rr, sm);
    rzn = block_sum(rzn, sm);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = rr;
        part_rz[blockIdx.x] = rzn;
    }
}
// the control block as the host will read it (pinned memory): written by the one thread that also writes the device copy
__device__ __forceinline__ void publish_ctrl(PcgCtrl* __restrict__ host_slot, int epoch, int done, int converged, int indef, int n_iter, double error)
{
    host_slot->converged = converged;
    host_slot->indef = indef;
    host_slot->error = error;
    __threadfence_system();
    host_slot->n_iter = n_iter;
    host_slot->done = done;
    __threadfence_system();
    host_slot->epoch = epoch;  // (the host looks at this first: written last)
    __threadfence_system();
}
// host_slot: non-null on the last iteration of a batch (the host looks at the control block there: no copy kernel, no extra boundary)
__global__ __launch_bounds__(BLOCK) void k_pcg_dir(int k, double abs_tol, double rel_tol, const double* __restrict__ part_rr, const double* __restrict__ part_rz, int nparts,
                                                   int64_t n, const double* __restrict__ z, double* __restrict__ p, PcgCtrl* __restrict__ ctrl, int stride,
                                                   PcgCtrl* __restrict__ host_slot, int epoch)
{
    const int done = ctrl->done;
    const double bb = ctrl->bb, rz_old = ctrl->rz[k & 1];
    // the first row's loads before the reduction (see k_pcg_step)
    const int64_t nrow = n / 3;
    int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, p0 = 0.0, p1 = 0.0, p2 = 0.0;
    if (row < nrow) {
        const size_t i = 3 * (size_t)row;
        z0 = z[i]; z1 = z[i + 1]; z2 = z[i + 2];
        p0 = p[i]; p1 = p[i + 1]; p2 = p[i + 2];
    }
    const bool scribe = blockIdx.x == 0 && threadIdx.x == 0;
    if (done == 1) {
        if (scribe && host_slot) publish_ctrl(host_slot, epoch, 1, ctrl->converged, ctrl->indef, ctrl->n_iter, ctrl->error);
        return;
    }
    if (done == 2) {  // indefiniteness stop decided in k_pcg_step of this iteration
        if (scribe) {
            ctrl->done = 1;
            if (host_slot) publish_ctrl(host_slot, epoch, 1, ctrl->converged, ctrl->indef, ctrl->n_iter, ctrl->error);
        }
        return;
    }
    __shared__ double sm[8];
    double rr, rz_new;
    sum_partials2(part_rr, part_rz, nparts, sm, stride, rr, rz_new);
    const double error = sqrt(rr / bb);
    const bool conv = error < abs_tol || error / 1.0 < rel_tol;  // error_0 = 1 for x0 = 0
    if (conv) {
        if (scribe) {
            ctrl->error = error;
            ctrl->n_iter = k;
            ctrl->converged = 1;
            ctrl->done = 1;
            if (host_slot) publish_ctrl(host_slot, epoch, 1, 1, ctrl->indef, k, error);
        }
        return;
    }
    const double beta = rz_new / rz_old;
    while (row < nrow) {
        const size_t i = 3 * (size_t)row;
        p[i] = z0 + beta * p0;
        p[i + 1] = z1 + beta * p1;
        p[i + 2] = z2 + beta * p2;
        row += (int64_t)gridDim.x * BLOCK;
        if (row < nrow) {
            const size_t j = 3 * (size_t)row;
            z0 = z[j]; z1 = z[j + 1]; z2 = z[j + 2];
            p0 = p[j]; p1 = p[j + 1]; p2 = p[j + 2];
        }
    }
    if (scribe) {
        ctrl->rz[(k + 1) & 1] = rz_new;
        ctrl->error = error;
        ctrl->n_iter = k;
        if (host_slot) publish_ctrl(host_slot, epoch, 0, 0, ctrl->indef, k, error);
    }
}

// ---- the same PCG on a row-sharded system (the three dot products of solve_pcg.h:180,201,217) ---------------------------------
// Every rank holds its block rows of A and the matching parts of x, r, z, q; p also carries the ghost columns. One iteration is TWO
// exchanges (all-gathers on the engine's stream) and five launches:
//   q = A p (ghosts of p are current), partial p.q | fold | all-gather of the ranks' p.q                                   [exchange 1: 8 bytes]
//   k_pcg_step with the sum (every rank adds the W numbers in rank order: the same bits everywhere): x, r, z; partial r.r, r.z
//   k_fold_pack: this rank's (r.r, r.z) and the z of the rows other ranks hold as ghosts, in one buffer | all-gather      [exchange 2]
//   k_pcg_dir_sharded: sums, convergence test, beta; p = z + beta p on the rank's rows AND on its ghosts (their z has just arrived, their
//   old p is here): the direction needs no exchange of its own
// The control block is computed redundantly and identically by every rank, so all of them stop at the same iteration; the host reads it
// every PCG_CHECK iterations. The solution is gathered into the global vector on every rank at the end.
__global__ __launch_bounds__(BLOCK) void k_fold_partials(const double* __restrict__ a, int na, const double* __restrict__ b, int nb, double* __restrict__ out)
{
    __shared__ double sm[4];
    const double sa = sum_partials(a, na, sm);
    const double sb = b ? sum_partials(b, nb, sm) : 0.0;
    if (threadIdx.x == 0) {
        out[0] = sa;
        if (b) out[1] = sb;
    }
}
// out = [sum a, sum b, z of the send rows (3 each)]: workgroup 0 folds, the others pack
__global__ __launch_bounds__(BLOCK) void k_fold_pack(const double* __restrict__ a, const double* __restrict__ b, int n, const double* __restrict__ z, const int32_t* __restrict__ send_rows,
                                                     int64_t n_send, double* __restrict__ out)
{
    if (blockIdx.x == 0) {
        __shared__ double sm[4];
        const double sa = sum_partials(a, n, sm);
        const double sb = sum_partials(b, n, sm);
        if (threadIdx.x == 0) {
            out[0] = sa;
            out[1] = sb;
        }
        return;
    }
    const int64_t t = (int64_t)(blockIdx.x - 1) * BLOCK + threadIdx.x;
    if (t >= 3 * n_send) return;
    const int64_t i = t / 3;
    out[2 + t] = z[3 * (int64_t)send_rows[i] + (t - 3 * i)];
}
// ghosts of p from the gathered buffer (stride S doubles per rank: two scalars, then the rank's send rows): p_ghost = z_ghost + beta p_ghost
__device__ __forceinline__ void ghosts_from_gathered(const double* __restrict__ recv, int64_t S, const int32_t* __restrict__ ghost_src, int64_t send_stride, int64_t n_ghost, int64_t n_own,
                                                     double beta, double* __restrict__ p)
{
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < 3 * n_ghost; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t g = t / 3, c = t - 3 * g;
        const int64_t src = ghost_src[g], o = src / send_stride, pos = src - o * send_stride;
        const double zg = recv[o * S + 2 + 3 * pos + c];
        double* pg = p + 3 * (n_own + g) + c;
        *pg = beta == 0.0 ? zg : zg + beta * *pg;
    }
}
__global__ __launch_bounds__(BLOCK) void k_pcg_init2_sharded(const double* __restrict__ recv, int W, int64_t S, double abs_tol, PcgCtrl* __restrict__ ctrl, const int32_t* __restrict__ ghost_src,
                                                             int64_t send_stride, int64_t n_ghost, int64_t n_own, double* __restrict__ p)
{
    double bb = 0.0, rz = 0.0;
    for (int r = 0; r < W; r++) {  // rank order: the same bits on every rank
        bb += recv[r * S];
        rz += recv[r * S + 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->bb = bb;
        ctrl->rz[1] = rz;
        ctrl->rz[0] = 0.0;
        ctrl->indef = 0;
        ctrl->n_iter = 0;
        ctrl->converged = 0;
        ctrl->done = 0;
        ctrl->error = 1.0;
        if (bb < abs_tol * abs_tol) {
            ctrl->done = 1;
            ctrl->converged = 1;
            ctrl->error = 0.0;
        } else if (1.0 < abs_tol) {
            ctrl->done = 1;
            ctrl->converged = 1;
        }
    }
    ghosts_from_gathered(recv, S, ghost_src, send_stride, n_ghost, n_own, 0.0, p);  // p_0 = z_0 on the ghosts too
}
__global__ __launch_bounds__(BLOCK) void k_pcg_dir_sharded(int k, double abs_tol, double rel_tol, const double* __restrict__ recv, int W, int64_t S, int64_t n, const double* __restrict__ z,
                                                           double* __restrict__ p, PcgCtrl* __restrict__ ctrl, const int32_t* __restrict__ ghost_src, int64_t send_stride, int64_t n_ghost,
                                                           int64_t n_own)
{
    const int done = ctrl->done;
    if (done == 1) return;
    if (done == 2) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->done = 1;
        return;
    }
    double rr = 0.0, rz_new = 0.0;
    for (int r = 0; r < W; r++) {
        rr += recv[r * S];
        rz_new += recv[r * S + 1];
    }
    const double error = sqrt(rr / ctrl->bb);
    const bool conv = error < abs_tol || error / 1.0 < rel_tol;
    if (conv) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctrl->error = error;
            ctrl->n_iter = k;
            ctrl->converged = 1;
            ctrl->done = 1;
        }
        return;
    }
    const double beta = rz_new / ctrl->rz[k & 1];
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) p[i] = z[i] + beta * p[i];
    ghosts_from_gathered(recv, S, ghost_src, send_stride, n_ghost, n_own, beta, p);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->rz[(k + 1) & 1] = rz_new;
        ctrl->error = error;
        ctrl->n_iter = k;
    }
}
static void pcg_sharded(Context& c, const double* rhs_global, double abs_tol, double rel_tol, int max_iter, int stop_on_indef, mistark_pcg_info* info)
{
    Shard& S = c.sh;
    const int64_t n_own = S.n_own;
    const int W = c.world;
    build_preconditioner(c);
    const int gv = grid_for(std::max<int64_t>(n_own, 1), BLOCK, VEC_GRID);
    BsrPart& m1 = c.part[1];
    const bool dyn = m1.nnzb > 0;
    double* part_pq = c.partials.p;
    double* part_rr = c.partials.p + MAX_PARTIALS;
    double* part_rz = c.partials.p + 2 * MAX_PARTIALS;
    double* part_bb = c.partials.p + 3 * MAX_PARTIALS;
    c.xl.ensure(3 * (size_t)std::max<int64_t>(S.n_loc, 1));
    const int64_t SS = 2 + 3 * S.send_stride;  // doubles per rank in the second exchange
    c.dist_scalar.ensure(8 + (size_t)W + (size_t)SS * (size_t)(W + 1));
    double* mine1 = c.dist_scalar.p;                    // [1]
    double* all1 = c.dist_scalar.p + 8;                 // [W]
    double* mine2 = c.dist_scalar.p + 8 + W;            // [SS]
    double* all2 = mine2 + SS;                          // [W * SS]
    if (rhs_global == c.tmp_b.p) throw Error("pcg: right-hand side in a scratch vector the sharded solve needs");
    double* b_l = c.tmp_b.p;  // local right-hand side
    shard_to_local(c, rhs_global, b_l, false);
    const int g_pack = 1 + grid_for(std::max<int64_t>(3 * S.n_send, 1));
    const int g_dir = std::max(gv, grid_for(std::max<int64_t>(3 * S.n_ghost, 1), BLOCK, VEC_GRID));
    if (3 * S.n_send < SS - 2) MS_CHECK(hipMemsetAsync(mine2, 0, (size_t)SS * sizeof(double), c.stream));  // (padding of the shorter send lists)
    hipLaunchKernelGGL(k_pcg_init, dim3(gv), dim3(BLOCK), 0, c.stream, (const double*)b_l, c.dinv.p, n_own, c.xl.p, c.r.p, c.z.p, c.p.p, part_bb, part_rz);
    hipLaunchKernelGGL(k_fold_pack, dim3(g_pack), dim3(BLOCK), 0, c.stream, (const double*)part_bb, (const double*)part_rz, gv, (const double*)c.z.p, (const int32_t*)S.send_rows.p, S.n_send, mine2);
    c.coll->allgather_f64(mine2, all2, (size_t)SS, c.stream);
    hipLaunchKernelGGL(k_pcg_init2_sharded, dim3(g_dir), dim3(BLOCK), 0, c.stream, (const double*)all2, W, SS, abs_tol, c.ctrl.p, (const int32_t*)S.ghost_src.p, S.send_stride, S.n_ghost, n_own,
                       c.p.p);
    constexpr int PCG_CHECK = 8;
    PcgCtrl h{};
    int k = 1;
    bool finished = false;
    std::vector<int> sampled_k, sampled_grid;
    while (!finished) {
        const int k_end = std::min(max_iter, k + PCG_CHECK - 1);
        for (; k <= k_end; k++) {
            // (SpMV timing for the bench's roofline figure, as in pcg(): one launch in 32 between a pair of events, an empty pair behind it)
            const bool sample = c.time_spmv && (k % 32) == 0 && sampled_k.size() < 64;
            if (sample) {
                while (c.ev.size() < 3 * (sampled_k.size() + 1)) {
                    hipEvent_t e;
                    MS_CHECK(hipEventCreate(&e));
                    c.ev.push_back(e);
                }
                MS_CHECK(hipEventRecord(c.ev[3 * sampled_k.size()], c.stream));
            }
            uint64_t* clk = nullptr;
            if (sample) {  // (and on the device clock, as in pcg(): per-workgroup start / end stamps in pinned memory)
                if (!c.spmv_clk_sharded) MS_CHECK(hipHostMalloc((void**)&c.spmv_clk_sharded, sizeof(uint64_t) * 64 * 2 * MAX_PARTIALS, hipHostMallocDefault));
                clk = c.spmv_clk_sharded + sampled_k.size() * 2 * MAX_PARTIALS;
                std::memset(clk, 0, sizeof(uint64_t) * 2 * MAX_PARTIALS);
            }
            const int gs = launch_spmv<0>(c, c.p.p, c.q.p, c.p.p, part_pq, c.ctrl.p, /*combine=*/false, clk);
            if (sample) {
                MS_CHECK(hipEventRecord(c.ev[3 * sampled_k.size() + 1], c.stream));
                MS_CHECK(hipEventRecord(c.ev[3 * sampled_k.size() + 2], c.stream));
                sampled_k.push_back(k);
                sampled_grid.push_back(gs);
            }
            hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(BLOCK), 0, c.stream, (const double*)part_pq, gs, (const double*)nullptr, 0, mine1);
            c.coll->allgather_f64(mine1, all1, 1, c.stream);
            hipLaunchKernelGGL(k_pcg_step, dim3(gv), dim3(BLOCK), 0, c.stream, k, stop_on_indef, (const double*)all1, W, c.dinv.p, n_own, c.p.p, c.q.p, c.xl.p, c.r.p, c.z.p, part_rr,
                               part_rz, c.ctrl.p, dyn ? (const int32_t*)m1.crow_of_row.p : nullptr, (const uint32_t*)m1.row_chunk0.p, (const double*)m1.yd.p,
                               (const double*)m1.chunk_partial.p);
            hipLaunchKernelGGL(k_fold_pack, dim3(g_pack), dim3(BLOCK), 0, c.stream, (const double*)part_rr, (const double*)part_rz, gv, (const double*)c.z.p, (const int32_t*)S.send_rows.p, S.n_send,
                               mine2);
            c.coll->allgather_f64(mine2, all2, (size_t)SS, c.stream);
            hipLaunchKernelGGL(k_pcg_dir_sharded, dim3(g_dir), dim3(BLOCK), 0, c.stream, k, abs_tol, rel_tol, (const double*)all2, W, SS, 3 * n_own, (const double*)c.z.p, c.p.p, c.ctrl.p,
                               (const int32_t*)S.ghost_src.p, S.send_stride, S.n_ghost, n_own);
        }
        fetch(c, &h, c.ctrl.p, sizeof(PcgCtrl));
        finished = h.done || k > max_iter;
    }
    for (size_t i = 0; i < sampled_k.size(); i++) {  // (the fetch above waited for the stream)
        if (h.done && sampled_k[i] > h.n_iter) continue;  // a no-op launch after convergence
        float ms = 0.f, ms_empty = 0.f;
        if (hipEventElapsedTime(&ms, c.ev[3 * i], c.ev[3 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms_empty, c.ev[3 * i + 1], c.ev[3 * i + 2]) == hipSuccess) {
            c.spmv_ms_sum += ms;
            c.spmv_empty_ms_sum += ms_empty;
            c.spmv_n++;
        }
        const uint64_t* clk = c.spmv_clk_sharded + i * 2 * MAX_PARTIALS;
        uint64_t t0 = ~0ull, t1 = 0;
        bool complete = true;
        for (int b = 0; b < sampled_grid[i]; b++) {
            if (clk[2 * b] == 0 || clk[2 * b + 1] == 0) { complete = false; break; }
            t0 = std::min(t0, clk[2 * b]);
            t1 = std::max(t1, clk[2 * b + 1]);
        }
        if (complete && t1 > t0) {
            c.spmv_clk_ticks += (double)(t1 - t0);
            c.spmv_clk_n++;
        }
    }
    shard_gather_global(c, c.xl.p, c.du.p);
    const int n_it = h.done ? h.n_iter : max_iter;
    c.last_cg_iters = n_it;
    if (info) {
        info->converged = h.done ? h.converged : 0;
        info->n_iterations = n_it;
        info->found_indefiniteness = h.indef;
        info->error = h.error;
        info->reserved = 0;
    }
}

// ---- the row-sharded PCG with ONE exposed exchange per iteration, for ranks that exchange through windows (dist.hpp: IpcView) --------------
// The five launches and two all-gathers of pcg_sharded become TWO launches whose workgroups push and poll the windows themselves. The
// arithmetic is the preconditioned CG of Chronopoulos & Gear (u = M^-1 r, w = A u, s = A p by recurrence), in which both dot products of
// an iteration are taken on the same vectors, so that p.Ap is not a reduction of its own (the three reductions of
// solve_pcg.h:180,201,217 are gamma = r.u, rr = r.r, and p.Ap = delta - beta gamma / alpha_prev with delta = w.u):
//   V_k  (k_cg_vec)    workgroup 0 first adds the rank's partial sums of iteration k-1 (r.u, r.r of V_{k-1}; w.u of S_{k-1}) and pushes the
//                      three numbers to every rank (message M2_{k-1}); then every workgroup adds, in rank order, the ranks' three numbers
//                      from its window: convergence test of iteration k-1, beta, p.Ap (indefiniteness test), alpha; p = u + beta p,
//                      s = w + beta s, x += alpha p, r -= alpha s, u = M^-1 r on its rows; partial (r.u, r.r) per workgroup to local memory;
//                      the new u of the rows other ranks reference as matrix columns is pushed to exactly those ranks           [message M1_k]
//   S_k  (k_spmv_halo) w = A u: columns of its own rows from memory, ghost columns straight from the window (the lane polls the granules of
//                      that ghost: rows without ghost columns never wait, so the halo hides behind the interior of the matrix); partial w.u per
//                      workgroup to local memory
// (History: version 1 let every workgroup of V and S push its partial sums to every rank and every workgroup of V add them all — thousands
// of uncached 8-byte reads per workgroup: 15 us per V launch at 43 k rows. Version 2 reduced them in a one-workgroup kernel R between S and V:
// V 7.6 us, R 3.0 us, a third launch. Version 3, this one: the reduction is workgroup 0 of V itself — the other workgroups poll for its push
// like for any other rank's; 6.4 + 7.8 us per rank and iteration at 8 ranks instead of 6.4 + 3.0 + 6.8.)
// The only wait that is not hidden is V_k's for the slowest rank's workgroup 0. Every rank adds the same numbers in the same order: identical
// bits, identical decisions, no all-reduce. Messages live in the fast region of the windows, two slots (parity of k) per message kind and
// source rank; a slot is rewritten two iterations later, when every reader has passed it (see "slot reuse" in dist.hip; between solves
// the all-gather of the solution separates the last readers from the next solve's first push).
struct CgFast
{
    IpcView v;
    size_t m1[2], m2[2];  // granule offset, inside every window, of rank 0's slot of the message kinds, per parity
    size_t m1_stride;     // granules per source rank in M1 (3 doubles per send row); M2 holds 3 doubles = 6 granules per rank
    int64_t send_stride;
};
constexpr size_t M2_STRIDE = 6;
// u of a send row to the ranks that hold it as a ghost
__device__ __forceinline__ void push_halo_row(const CgFast& f, int par, uint32_t tag, int sp, uint32_t mask, double u0, double u1, double u2)
{
    const size_t at = f.m1[par] + (size_t)f.v.rank * f.m1_stride + 6 * (size_t)sp;
    while (mask) {
        const int q = __ffs(mask) - 1;
        mask &= mask - 1;
        unsigned long long* g = f.v.win[q] + at;
        granule_store_f64(g, tag, u0);
        granule_store_f64(g + 2, tag, u1);
        granule_store_f64(g + 4, tag, u2);
    }
}
// prologue: x = 0, r = b, u = M^-1 r (the preconditioner is built); p = s = 0; control block; halo of u (message M1_0)
__global__ __launch_bounds__(BLOCK) void k_cg_prologue(CgFast f, uint32_t tag_out, const double* __restrict__ b, const float* __restrict__ dinv, int64_t n_own, double* __restrict__ x,
                                                       double* __restrict__ r, double* __restrict__ u, double* __restrict__ p, double* __restrict__ s, PcgCtrl* __restrict__ ctrl,
                                                       const int32_t* __restrict__ send_pos_of_row, const uint32_t* __restrict__ send_mask, double* __restrict__ part_ru,
                                                       double* __restrict__ part_rr)
{
    __shared__ double sm[4];
    double bb = 0.0, ru = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < n_own; row += (int64_t)gridDim.x * BLOCK) {
        const size_t i = 3 * (size_t)row;
        const double r0 = b[i], r1 = b[i + 1], r2 = b[i + 2];
        const float* d = dinv + 9 * row;
        const double u0 = (double)d[0] * r0 + (double)d[1] * r1 + (double)d[2] * r2;
        const double u1 = (double)d[3] * r0 + (double)d[4] * r1 + (double)d[5] * r2;
        const double u2 = (double)d[6] * r0 + (double)d[7] * r1 + (double)d[8] * r2;
        x[i] = 0.0; x[i + 1] = 0.0; x[i + 2] = 0.0;
        p[i] = 0.0; p[i + 1] = 0.0; p[i + 2] = 0.0;
        s[i] = 0.0; s[i + 1] = 0.0; s[i + 2] = 0.0;
        r[i] = r0; r[i + 1] = r1; r[i + 2] = r2;
        u[i] = u0; u[i + 1] = u1; u[i + 2] = u2;
        bb += r0 * r0 + r1 * r1 + r2 * r2;
        ru += r0 * u0 + r1 * u1 + r2 * u2;
        const int sp = send_pos_of_row[row];
        if (sp >= 0) push_halo_row(f, 0, tag_out, sp, send_mask[sp], u0, u1, u2);
    }
    bb = block_sum(bb, sm);
    ru = block_sum(ru, sm);
    if (threadIdx.x == 0) {
        part_ru[blockIdx.x] = ru;
        part_rr[blockIdx.x] = bb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->bb = 0.0;
        ctrl->rz[0] = ctrl->rz[1] = 0.0;
        ctrl->alpha[0] = ctrl->alpha[1] = 0.0;
        ctrl->indef = 0;
        ctrl->n_iter = 0;
        ctrl->converged = 0;
        ctrl->done = 0;
        ctrl->error = 1.0;
    }
}
struct VecRow
{
    double u0, u1, u2, w0, w1, w2, p0, p1, p2, s0, s1, s2, x0, x1, x2, r0, r1, r2;
    float d[9];
    int sp;
};
__device__ __forceinline__ void vec_load(VecRow& v, int64_t row, const float* __restrict__ dinv, const double* __restrict__ u, const double* __restrict__ w, const double* __restrict__ p,
                                         const double* __restrict__ s, const double* __restrict__ x, const double* __restrict__ r, const int32_t* __restrict__ crow_of_row,
                                         const uint32_t* __restrict__ row_chunk0, const double* __restrict__ yd, const double* __restrict__ chunk_partial,
                                         const int32_t* __restrict__ send_pos_of_row)
{
    const size_t i = 3 * (size_t)row;
    v.u0 = u[i]; v.u1 = u[i + 1]; v.u2 = u[i + 2];
    v.w0 = w[i]; v.w1 = w[i + 1]; v.w2 = w[i + 2];
    v.p0 = p[i]; v.p1 = p[i + 1]; v.p2 = p[i + 2];
    v.s0 = s[i]; v.s1 = s[i + 1]; v.s2 = s[i + 2];
    v.x0 = x[i]; v.x1 = x[i + 1]; v.x2 = x[i + 2];
    v.r0 = r[i]; v.r1 = r[i + 1]; v.r2 = r[i + 2];
#pragma unroll
    for (int k = 0; k < 9; k++) v.d[k] = dinv[9 * row + k];
    v.sp = send_pos_of_row ? send_pos_of_row[row] : -1;
    // (+ the contact part of w, which the SpMV left in yd / chunk_partial: dyn_row_wave, called by the whole wavefront behind this)
}
// V_k, k >= 1 (check_only: the convergence test of iteration k - 1 and nothing else, behind the last iteration the caller allows).
// replay (mistark_dist_fused_bench): the kernel of a FINISHED solve launched again on the messages still in the window — every poll is
// answered at once, no decision is taken, the control block stays as it is: the kernel's own duration.
__global__ __launch_bounds__(BLOCK) void k_cg_vec(int k, int check_only, int stop_on_indef, double abs_tol, double rel_tol, CgFast f, uint32_t tag_m2_in, uint32_t tag_out,
                                                  const float* __restrict__ dinv, int64_t n_own, double* __restrict__ u, const double* __restrict__ w, double* __restrict__ p,
                                                  double* __restrict__ s, double* __restrict__ x, double* __restrict__ r, PcgCtrl* __restrict__ ctrl,
                                                  const int32_t* __restrict__ crow_of_row, const uint32_t* __restrict__ row_chunk0, const double* __restrict__ yd,
                                                  const double* __restrict__ chunk_partial, const int32_t* __restrict__ send_pos_of_row, const uint32_t* __restrict__ send_mask,
                                                  double* __restrict__ part_ru, double* __restrict__ part_rr, PcgCtrl* __restrict__ host_slot, int epoch, int replay,
                                                  const double* __restrict__ loc_wu, int loc_gs, const double* __restrict__ loc_ru, const double* __restrict__ loc_rr, int loc_gv,
                                                  int windows)
{
    // loc_*: the partial sums the previous SpMV (w.u) and vector kernel (r.u, r.r; the other parity's buffers than the ones this launch
    // writes) left in local memory. ONE GPU (pcg_cg, windows == 0): every workgroup re-reduces them, no pushes. Ranks on windows: workgroup
    // 0 reduces them and pushes the rank's three sums to every rank (message M2_{k-1}) before it polls like the others
    const bool scribe = blockIdx.x == 0 && threadIdx.x == 0 && !replay;
    if (!replay && ctrl->done) {
        if (scribe && host_slot) publish_ctrl(host_slot, epoch, 1, ctrl->converged, ctrl->indef, ctrl->n_iter, ctrl->error);
        return;
    }
    const int i = k - 1, par_in = i & 1, par_out = k & 1;
    int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    VecRow v;
    // (the thread's row is requested before the sums below: they wait for the slowest rank's reduction)
    if (!check_only && row < n_own) vec_load(v, row, dinv, u, w, p, s, x, r, crow_of_row, row_chunk0, yd, chunk_partial, send_pos_of_row);
    if (!check_only && crow_of_row) dyn_row_wave(crow_of_row, row_chunk0, yd, chunk_partial, row < n_own ? row : -1, v.w0, v.w1, v.w2);
    __shared__ double sm[3 * MAX_IPC_RANKS + 8];
    double gamma = 0.0, rr = 0.0, delta = 0.0;
    if (!windows) {
        sum_partials2(loc_ru, loc_rr, loc_gv, sm, 1, gamma, rr);
        __syncthreads();
        delta = sum_partials(loc_wu, loc_gs, sm);
        __syncthreads();
    } else {
        const int W = f.v.world;
        if (blockIdx.x == 0) {  // (uniform per workgroup)
            double g1, r1;
            sum_partials2(loc_ru, loc_rr, loc_gv, sm, 1, g1, r1);
            __syncthreads();
            const double d1 = sum_partials(loc_wu, loc_gs, sm);
            __syncthreads();
            if (threadIdx.x < (unsigned)W) {
                unsigned long long* g = f.v.win[threadIdx.x] + f.m2[par_in] + (size_t)f.v.rank * M2_STRIDE;
                granule_store_f64(g, tag_m2_in, g1);
                granule_store_f64(g + 2, tag_m2_in, r1);
                granule_store_f64(g + 4, tag_m2_in, d1);
            }
        }
        if (threadIdx.x < (unsigned)(3 * W)) {  // one lane per (rank, component); added below in rank order
            const unsigned long long* g = f.v.win[f.v.rank] + f.m2[par_in] + 2 * (size_t)threadIdx.x;  // (rank-major: 6 granules per rank)
            sm[threadIdx.x] = granule_wait_f64(g, tag_m2_in, f.v.err, wall_clock64(), f.v.timeout_ticks, 2u | ((unsigned)k << 8));
        }
        __syncthreads();
        for (int q = 0; q < W; q++) {
            gamma += sm[3 * q];
            rr += sm[3 * q + 1];
            delta += sm[3 * q + 2];
        }
    }
    double error = 1.0;
    if (replay) {
        // (no exits)
    } else if (i == 0) {  // rr = b.b: the two exits before the first iteration (solve_pcg.h:125-131,150-156)
        const bool zero_rhs = rr < abs_tol * abs_tol;
        if (zero_rhs || 1.0 < abs_tol) {
            if (scribe) {
                ctrl->bb = rr;
                ctrl->error = zero_rhs ? 0.0 : 1.0;
                ctrl->n_iter = 0;
                ctrl->converged = 1;
                ctrl->done = 1;
                if (host_slot) publish_ctrl(host_slot, epoch, 1, 1, 0, 0, zero_rhs ? 0.0 : 1.0);
            }
            return;
        }
    } else {
        error = sqrt(rr / ctrl->bb);
        if (error < abs_tol || error / 1.0 < rel_tol) {  // error_0 = 1 for x0 = 0
            if (scribe) {
                ctrl->error = error;
                ctrl->n_iter = i;
                ctrl->converged = 1;
                ctrl->done = 1;
                if (host_slot) publish_ctrl(host_slot, epoch, 1, 1, ctrl->indef, i, error);
            }
            return;
        }
    }
    if (check_only) {
        if (scribe) {
            ctrl->error = error;
            ctrl->n_iter = i;
            if (host_slot) publish_ctrl(host_slot, epoch, 0, 0, ctrl->indef, k, error);
        }
        return;
    }
    double beta = 0.0, pAp = delta;
    if (i > 0) {
        beta = gamma / ctrl->rz[(i - 1) & 1];
        pAp = delta - beta * gamma / ctrl->alpha[(i - 1) & 1];
    }
    if (pAp <= 0.0 && !replay) {  // solve_pcg.h:183-190
        if (stop_on_indef) {
            if (scribe) {
                ctrl->indef = 1;
                ctrl->n_iter = k;
                ctrl->converged = 0;
                ctrl->done = 1;
                if (host_slot) publish_ctrl(host_slot, epoch, 1, 0, 1, k, error);
            }
            return;  // every workgroup of every rank takes the same decision: x stays untouched
        }
        if (scribe) ctrl->indef = 1;
    }
    const double alpha = gamma / pAp;
    double ru = 0.0, rrn = 0.0;
    while (__any(row < n_own)) {  // (wave-uniform: dyn_row_wave needs the whole wavefront)
      if (row < n_own) {
        const size_t j = 3 * (size_t)row;
        const double p0 = beta == 0.0 ? v.u0 : v.u0 + beta * v.p0, p1 = beta == 0.0 ? v.u1 : v.u1 + beta * v.p1, p2 = beta == 0.0 ? v.u2 : v.u2 + beta * v.p2;
        const double s0 = beta == 0.0 ? v.w0 : v.w0 + beta * v.s0, s1 = beta == 0.0 ? v.w1 : v.w1 + beta * v.s1, s2 = beta == 0.0 ? v.w2 : v.w2 + beta * v.s2;
        const double r0 = v.r0 - alpha * s0, r1 = v.r1 - alpha * s1, r2 = v.r2 - alpha * s2;
        p[j] = p0; p[j + 1] = p1; p[j + 2] = p2;
        s[j] = s0; s[j + 1] = s1; s[j + 2] = s2;
        x[j] = v.x0 + alpha * p0; x[j + 1] = v.x1 + alpha * p1; x[j + 2] = v.x2 + alpha * p2;
        r[j] = r0; r[j + 1] = r1; r[j + 2] = r2;
        const float* d = v.d;
        const double u0 = (double)d[0] * r0 + (double)d[1] * r1 + (double)d[2] * r2;
        const double u1 = (double)d[3] * r0 + (double)d[4] * r1 + (double)d[5] * r2;
        const double u2 = (double)d[6] * r0 + (double)d[7] * r1 + (double)d[8] * r2;
        u[j] = u0; u[j + 1] = u1; u[j + 2] = u2;
        rrn += r0 * r0 + r1 * r1 + r2 * r2;
        ru += r0 * u0 + r1 * u1 + r2 * u2;
        if (v.sp >= 0) push_halo_row(f, par_out, tag_out, v.sp, send_mask[v.sp], u0, u1, u2);
        row += (int64_t)gridDim.x * BLOCK;
        if (row < n_own) vec_load(v, row, dinv, u, w, p, s, x, r, crow_of_row, row_chunk0, yd, chunk_partial, send_pos_of_row);
      }
      if (crow_of_row && __any(row < n_own)) dyn_row_wave(crow_of_row, row_chunk0, yd, chunk_partial, row < n_own ? row : -1, v.w0, v.w1, v.w2);
    }
    __syncthreads();
    rrn = block_sum(rrn, sm);
    ru = block_sum(ru, sm);
    if (threadIdx.x == 0) {
        part_ru[blockIdx.x] = ru;
        part_rr[blockIdx.x] = rrn;
    }
    if (scribe) {
        if (i == 0) ctrl->bb = rr;
        ctrl->rz[i & 1] = gamma;
        ctrl->alpha[i & 1] = alpha;
        ctrl->error = error;
        ctrl->n_iter = i;
        if (host_slot) publish_ctrl(host_slot, epoch, 0, 0, ctrl->indef, k, error);
    }
}
// x of the SpMV for k_spmv_halo: own columns from memory, ghost columns from the window (M1 of this iteration), polled by the lane that needs them
struct XHalo
{
    const double* x;                // u, own rows
    const unsigned long long* mine; // own window
    size_t halo0;                   // granule offset of rank 0's halo values (the M1 slot of this parity)
    size_t m1_stride;
    const int32_t* ghost_src;       // per ghost: owner * send_stride + position among the owner's send rows
    int64_t send_stride;
    size_t own3;                    // 3 * n_own
    uint32_t tag;
    unsigned int* err;
    unsigned long long t0, budget;
    unsigned int code;              // which wait this is, for the error message (5 | iteration << 8)
    __device__ __forceinline__ void load(size_t c3, double& x0, double& x1, double& x2) const
    {
        if (c3 < own3) {
            x0 = x[c3];
            x1 = x[c3 + 1];
            x2 = x[c3 + 2];
        } else {
            const int64_t src = ghost_src[(c3 - own3) / 3], o = src / send_stride, pos = src - o * send_stride;
            const unsigned long long* g = mine + halo0 + (size_t)o * m1_stride + 6 * (size_t)pos;
            x0 = granule_wait_f64(g, tag, err, t0, budget, code);
            x1 = granule_wait_f64(g + 2, tag, err, t0, budget, code);
            x2 = granule_wait_f64(g + 4, tag, err, t0, budget, code);
        }
    }
    __device__ __forceinline__ bool has_dot() const { return true; }
    __device__ __forceinline__ double row_dot(size_t r3, double y0, double y1, double y2) const { return x[r3] * y0 + x[r3 + 1] * y1 + x[r3 + 2] * y2; }
    __device__ __forceinline__ void row_pre(size_t r3, double& p0, double& p1, double& p2) const
    {
        p0 = x[r3];
        p1 = x[r3 + 1];
        p2 = x[r3 + 2];
    }
    __device__ __forceinline__ double row_dot_pre(size_t, double p0, double p1, double p2, double y0, double y1, double y2) const { return p0 * y0 + p1 * y1 + p2 * y2; }
};
__device__ __forceinline__ double row_dot_nostore(const XHalo& X, size_t r3, double y0, double y1, double y2) { return X.row_dot(r3, y0, y1, y2); }
// S_k: w = A u (+ the contact part's row sums, as k_spmv_fused leaves them); the workgroups' partial sums of w.u stay in local memory
__global__ __launch_bounds__(BLOCK) void k_spmv_halo(int g0, int gr, int g1, StaticPart m, DynPart d, XHalo X, double* __restrict__ y, double* __restrict__ partials,
                                                    const PcgCtrl* __restrict__ ctrl, uint64_t* __restrict__ clk, int replay)
{
    if (!replay && ctrl->done) return;
    const int b = (int)blockIdx.x;
    const uint64_t t_start = wall_clock64();
    X.t0 = t_start;
    if (b < g1) spmv_chunks(b, g1, d, X, partials + g0 + gr);
    else if (b < g1 + gr) spmv_long_rows(b - g1, gr, m.vals, m.scol, m.long_rows, m.n_long_rows, m.row_ptr, m.row_pos, X, y, partials + g0);
    else spmv_chunked_static<0>(b - g1 - gr, g0, m.vals, m.scol, m.tile_first_row, m.n_chunks, m.chunk_tiles, X, y, partials);
    if (clk) {
        __syncthreads();
        if (threadIdx.x == 0) {
            clk[2 * b] = t_start;
            clk[2 * b + 1] = wall_clock64();
        }
    }
}
static double now_seconds();
namespace {
// want[owner * send_stride + position] = 1 for every ghost column the matrix references
__global__ __launch_bounds__(BLOCK) void k_mark_ghost_refs(const uint32_t* __restrict__ colw, int64_t n, int64_t n_own, const int32_t* __restrict__ ghost_src, double* __restrict__ want)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t col = (int64_t)(colw[i] & 0x7fffffffu);
    if (col >= n_own) want[ghost_src[col - n_own]] = 1.0;
}
// bit q of mask[pos]: rank q references my send row `pos` (all[q] is rank q's want table)
__global__ __launch_bounds__(BLOCK) void k_build_send_mask(const double* __restrict__ all, int W, int me, int64_t send_stride, int64_t n_send, const uint32_t* __restrict__ holders,
                                                          uint32_t* __restrict__ mask)
{
    const int64_t pos = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= n_send) return;
    uint32_t m = 0;
    for (int q = 0; q < W; q++)
        if (q != me && all[(size_t)q * (size_t)W * (size_t)send_stride + (size_t)me * (size_t)send_stride + (size_t)pos] != 0.0) m |= 1u << q;
    mask[pos] = m & holders[pos];
}
// The halo of the fused iteration goes only to the ranks whose matrix has the row as a column: every rank marks the ghosts its two matrix
// parts reference, one all-gather carries the marks to the owners. Again whenever a pattern or the element lists changed (collective: every
// rank builds its patterns at the same points of the same control flow).
void fused_refresh_masks(Context& c)
{
    Shard& S = c.sh;
    if (c.cg_mask_pattern == c.pattern_version && c.cg_mask_lists == S.version_lists) return;
    const int W = c.world;
    const size_t n = (size_t)W * (size_t)std::max<int64_t>(S.send_stride, 1);
    c.cg_want_s.ensure(n);
    c.cg_want_r.ensure(n * (size_t)W);
    c.cg_send_mask.ensure((size_t)std::max<int64_t>(S.n_send, 1));
    MS_CHECK(hipMemsetAsync(c.cg_want_s.p, 0, n * sizeof(double), c.stream));
    if (S.n_ghost > 0) {
        const BsrPart& m0 = c.part[0];
        const BsrPart& m1 = c.part[1];
        if (m0.ntiles > 0)
            hipLaunchKernelGGL(k_mark_ghost_refs, dim3(grid_for(m0.ntiles * 64)), dim3(BLOCK), 0, c.stream, (const uint32_t*)m0.scol.p, m0.ntiles * 64, S.n_own, (const int32_t*)S.ghost_src.p,
                               c.cg_want_s.p);
        if (m1.nnzb > 0)
            hipLaunchKernelGGL(k_mark_ghost_refs, dim3(grid_for(m1.nnzb)), dim3(BLOCK), 0, c.stream, (const uint32_t*)m1.colw.p, m1.nnzb, S.n_own, (const int32_t*)S.ghost_src.p, c.cg_want_s.p);
    }
    c.coll->allgather_f64(c.cg_want_s.p, c.cg_want_r.p, n, c.stream);
    if (S.n_send > 0)
        hipLaunchKernelGGL(k_build_send_mask, dim3(grid_for(S.n_send)), dim3(BLOCK), 0, c.stream, (const double*)c.cg_want_r.p, W, c.rank, std::max<int64_t>(S.send_stride, 1), S.n_send,
                           (const uint32_t*)S.send_mask.p, c.cg_send_mask.p);
    c.cg_mask_pattern = c.pattern_version;
    c.cg_mask_lists = S.version_lists;
}
// everything the launches of one fused solve share
struct FusedSolve
{
    Context& c;
    CgFast f;
    int g0, gr, g1, gs, gv;
    StaticPart sp;
    DynPart d;
    uint32_t base;
    double *u, *w, *p, *s, *x, *r, *part_wu;
    double* pr[2][2];  // partial (r.u, r.r) of the vector kernels by parity of k (V_k reads V_{k-1}'s while it writes its own)
    const uint32_t* send_mask;  // where the halo goes: the ranks that reference the row (fused_refresh_masks), or every holder
    uint32_t tag_m1(int i) const { return base + 2u * (uint32_t)i + 1u; }
    uint32_t tag_m2(int i) const { return base + 2u * (uint32_t)i + 2u; }
    void launch_S(int i, uint64_t* clk, int replay) const
    {
        const Shard& S = c.sh;
        XHalo X{};
        X.x = u;
        X.mine = f.v.win[f.v.rank];
        X.halo0 = f.m1[i & 1];
        X.m1_stride = f.m1_stride;
        X.ghost_src = S.ghost_src.p;
        X.send_stride = std::max<int64_t>(S.send_stride, 1);
        X.own3 = 3 * (size_t)S.n_own;
        X.tag = tag_m1(i);
        X.err = f.v.err;
        X.budget = f.v.timeout_ticks;
        X.code = 5u | ((unsigned)i << 8);
        hipLaunchKernelGGL(k_spmv_halo, dim3(gs), dim3(BLOCK), 0, c.stream, g0, gr, g1, sp, d, X, w, part_wu, (const PcgCtrl*)c.ctrl.p, clk, replay);
    }
    void launch_V(int k, bool check_only, int stop_on_indef, double abs_tol, double rel_tol, PcgCtrl* host_slot, int epoch, int replay) const
    {
        const Shard& S = c.sh;
        BsrPart& m1 = c.part[1];
        const bool dyn = m1.nnzb > 0;
        hipLaunchKernelGGL(k_cg_vec, dim3(gv), dim3(BLOCK), 0, c.stream, k, check_only ? 1 : 0, stop_on_indef, abs_tol, rel_tol, f, tag_m2(k - 1), tag_m1(k), (const float*)c.dinv.p, S.n_own,
                           u, (const double*)w, p, s, x, r, c.ctrl.p, dyn ? (const int32_t*)m1.crow_of_row.p : (const int32_t*)nullptr, (const uint32_t*)m1.row_chunk0.p,
                           (const double*)m1.yd.p, (const double*)m1.chunk_partial.p, (const int32_t*)S.send_pos_of_row.p, send_mask, pr[k & 1][0], pr[k & 1][1], host_slot,
                           epoch, replay, (const double*)part_wu, gs, (const double*)pr[(k - 1) & 1][0], (const double*)pr[(k - 1) & 1][1], gv, 1);
    }
};
// false: no windows, too many ranks, or the halo does not fit the fast region
bool fused_setup(Context& c, FusedSolve& F)
{
    const IpcView* view = c.coll ? c.coll->ipc() : nullptr;
    if (!view || c.no_fused_pcg || c.world > MAX_IPC_RANKS) return false;
    Shard& S = c.sh;
    const int W = c.world;
    spmv_launch_shape(c, F.g0, F.gr, F.g1, F.sp, F.d);
    F.gs = F.g0 + F.gr + F.g1;
    F.gv = grid_for(std::max<int64_t>(S.n_own, 1), BLOCK, PCG_GRID);
    // (ranks sharing ONE device — test boxes —: every rank's polling workgroups must leave room for the kernels they wait for; the same cap as the
    // SpMV's. The vector kernel walks its rows with a grid stride, any grid is correct.)
    if (c.spmv_grid_cap > 0) F.gv = std::min(F.gv, std::max(c.spmv_grid_cap / 2, 8));
    F.f = CgFast{};
    F.f.v = *view;
    F.f.send_stride = S.send_stride;
    F.f.m1_stride = 6 * (size_t)std::max<int64_t>(S.send_stride, 1);
    const size_t per_parity = (size_t)W * (F.f.m1_stride + M2_STRIDE);
    if (2 * per_parity > view->fast_granules) return false;
    for (int par = 0; par < 2; par++) {
        F.f.m1[par] = view->fast_off + (size_t)par * per_parity;
        F.f.m2[par] = F.f.m1[par] + (size_t)W * F.f.m1_stride;
    }
    c.xl.ensure(3 * (size_t)std::max<int64_t>(S.n_loc, 1));
    c.p2.ensure(3 * (size_t)std::max<int64_t>(S.n_loc, 1));
    F.u = c.z.p;
    F.w = c.q.p;
    F.p = c.p.p;
    F.s = c.p2.p;
    F.x = c.xl.p;
    F.r = c.r.p;
    F.part_wu = c.partials.p;
    F.pr[0][0] = c.partials.p + 4 * MAX_PARTIALS;
    F.pr[0][1] = c.partials.p + 5 * MAX_PARTIALS;
    F.pr[1][0] = c.partials.p + 2 * MAX_PARTIALS;
    F.pr[1][1] = c.partials.p + 3 * MAX_PARTIALS;
    if (view->fast_tag) c.fused_tag = std::max(c.fused_tag, *view->fast_tag);  // (an earlier context on the same windows: continue behind its tags)
    F.base = c.fused_tag;
    F.send_mask = c.no_halo_subset ? (const uint32_t*)S.send_mask.p : (const uint32_t*)c.cg_send_mask.p;
    return true;
}
}  // namespace
// false: this solve cannot take the fused iteration
static bool pcg_sharded_fused(Context& c, const double* rhs_global, double abs_tol, double rel_tol, int max_iter, int stop_on_indef, mistark_pcg_info* info)
{
    FusedSolve F{c};
    if (!fused_setup(c, F)) return false;
    Shard& S = c.sh;
    const int me = c.rank;
    if (!c.no_halo_subset) {
        fused_refresh_masks(c);
        F.send_mask = c.cg_send_mask.p;  // (the buffer may have been allocated just now)
    }
    build_preconditioner(c);
    static const bool dbg = std::getenv("MISTARK_DEBUG_FUSED") != nullptr;
    if (dbg)
        std::fprintf(stderr, "[fused r%d] gv=%d gs=%d (g0 %d gr %d g1 %d) n_own=%lld send_stride=%lld tag base %u max_iter %d\n", me, F.gv, F.gs, F.g0, F.gr, F.g1, (long long)S.n_own,
                     (long long)S.send_stride, c.fused_tag, max_iter);
    if (rhs_global == c.tmp_b.p) throw Error("pcg: right-hand side in a scratch vector the sharded solve needs");
    double* b_l = c.tmp_b.p;
    shard_to_local(c, rhs_global, b_l, false);
    hipLaunchKernelGGL(k_cg_prologue, dim3(F.gv), dim3(BLOCK), 0, c.stream, F.f, F.tag_m1(0), (const double*)b_l, (const float*)c.dinv.p, S.n_own, F.x, F.r, F.u, F.p, F.s, c.ctrl.p,
                       (const int32_t*)S.send_pos_of_row.p, F.send_mask, F.pr[0][0], F.pr[0][1]);
    std::vector<int> sampled_i;
    auto launch_S = [&](int i) {
        uint64_t* clk = nullptr;
        if (c.time_spmv && i > 0 && (i % 32) == 0 && sampled_i.size() < 64) {  // (device-clock sample for the bench's roofline figure, as in pcg())
            if (!c.spmv_clk_sharded) MS_CHECK(hipHostMalloc((void**)&c.spmv_clk_sharded, sizeof(uint64_t) * 64 * 2 * MAX_PARTIALS, hipHostMallocDefault));
            clk = c.spmv_clk_sharded + sampled_i.size() * 2 * MAX_PARTIALS;
            std::memset(clk, 0, sizeof(uint64_t) * 2 * MAX_PARTIALS);
            sampled_i.push_back(i);
        }
        F.launch_S(i, clk, 0);
    };
    // batches of [S_{k-1}, R_{k-1}, V_k] with one look-ahead batch in flight, as in pcg(): the last V of a batch writes the control block to a
    // pinned slot the host watches
    constexpr int BATCH = 8;
    PcgCtrl* hs[2] = {reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048), reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048 + 64)};
    const int epoch = ++c.pcg_epoch;
    int k = 1;  // next V to launch
    bool tail_done = false;  // the check-only V behind iteration max_iter has been launched
    auto launch_batch = [&](int slot) {
        hs[slot]->epoch = epoch - 1;
        hs[slot]->done = 0;
        hs[slot]->n_iter = -1;
        const int k_end = std::min(max_iter + 1, k + BATCH - 1);
        for (; k <= k_end; k++) {
            const bool check_only = k == max_iter + 1;
            launch_S(k - 1);      // w_{k-1}
            F.launch_V(k, check_only, stop_on_indef, abs_tol, rel_tol, k == k_end ? hs[slot] : (PcgCtrl*)nullptr, epoch, 0);
            if (check_only) tail_done = true;
        }
        return k_end;
    };
    PcgCtrl h{};
    int slot = 0;
    int k_end_cur = launch_batch(0);
    for (;;) {
        const bool more = !tail_done;
        int k_end_next = 0;
        if (more) k_end_next = launch_batch(slot ^ 1);
        const volatile PcgCtrl* v = hs[slot];
        const double t_wait = now_seconds();
        auto reported = [&] { return v->epoch == epoch && (v->done || v->n_iter >= k_end_cur); };
        for (uint64_t spins = 0; !reported(); spins++) {
            __builtin_ia32_pause();
            if ((spins & 0xfffff) != 0xfffff) continue;
            c.coll->check();
            const hipError_t q = hipStreamQuery(c.stream);
            if (q != hipErrorNotReady) {
                MS_CHECK(q);
                if (!reported()) {
                    PcgCtrl dev{};
                    MS_CHECK(hipMemcpy(&dev, c.ctrl.p, sizeof(PcgCtrl), hipMemcpyDeviceToHost));
                    hs[slot]->converged = dev.converged;
                    hs[slot]->indef = dev.indef;
                    hs[slot]->error = dev.error;
                    hs[slot]->n_iter = dev.done ? dev.n_iter : k_end_cur;
                    hs[slot]->done = dev.done ? 1 : 0;
                    hs[slot]->epoch = epoch;
                }
                break;
            }
            if (now_seconds() - t_wait > 120.0) throw Error("sharded pcg: the device did not report iteration " + std::to_string(k_end_cur) + " within 120 s");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        h = *hs[slot];
        if (dbg) std::fprintf(stderr, "[fused r%d] batch to %d: done %d conv %d indef %d n_iter %d err %g (k next %d)\n", me, k_end_cur, h.done, h.converged, h.indef, h.n_iter, h.error, k);
        if (h.done || !more) break;
        slot ^= 1;
        k_end_cur = k_end_next;
    }
    // the next solve's tags start behind the last one any rank can have used in this one (a rank launches at most two batches beyond the
    // iteration that ended the solve; computed from the iteration count, which is the same number on every rank)
    c.fused_tag = F.base + 2u * (uint32_t)((h.done ? h.n_iter : max_iter) + 2 * BATCH + 4);
    if (const IpcView* v = c.coll->ipc())
        if (v->fast_tag) *v->fast_tag = c.fused_tag;
    if (c.fused_tag > 0xf0000000u) throw Error("sharded PCG: the window tags are about to wrap to the windows' zero-filled state after ~2^32 exchanges; create a new communicator");
    shard_gather_global(c, F.x, c.du.p);  // (also the barrier between this solve's last window readers and the next solve's first push)
    MS_CHECK(hipStreamSynchronize(c.stream));
    c.coll->check();
    if (c.time_spmv) {
        for (size_t q = 0; q < sampled_i.size(); q++) {
            if (h.done && sampled_i[q] >= h.n_iter) continue;  // (a no-op launch after the solve was over)
            const uint64_t* clk = c.spmv_clk_sharded + q * 2 * MAX_PARTIALS;
            uint64_t t0 = ~0ull, t1 = 0;
            bool complete = true;
            for (int b = 0; b < F.gs; b++) {
                if (clk[2 * b] == 0 || clk[2 * b + 1] == 0) { complete = false; break; }
                t0 = std::min(t0, clk[2 * b]);
                t1 = std::max(t1, clk[2 * b + 1]);
            }
            if (complete && t1 > t0) {
                c.spmv_clk_ticks += (double)(t1 - t0);
                c.spmv_clk_n++;
            }
        }
    }
    const int n_it = h.done ? h.n_iter : max_iter;
    c.last_cg_iters = n_it;
    if (info) {
        info->converged = h.done ? h.converged : 0;
        info->n_iterations = n_it;
        info->found_indefiniteness = h.indef;
        info->error = h.error;
        info->reserved = 0;
    }
    // what mistark_dist_fused_bench replays: S_n, R_n and V_{n+1} of a converged solve found the messages M1_n / M2_n complete, and nobody has
    // pushed behind them
    c.fused_replay.valid = h.done && h.converged && !h.indef && n_it >= 1;
    c.fused_replay.base = F.base;
    c.fused_replay.n = n_it;
    c.fused_replay.pattern = c.pattern_version;
    return true;
}
// Solo durations of the two kernels of the fused iteration on this rank's shard: n launches each of S_n and V_{n+1} of the last
// converged solve, back to back (see `replay` in k_cg_vec), between HIP events. NO other rank may start a solve meanwhile (the caller takes
// turns: mistark_dist_fused_bench).
void fused_pcg_replay(Context& c, int n_launches, double* s_us, double* v_us)
{
    if (!c.fused_replay.valid || c.fused_replay.pattern != c.pattern_version) throw Error("fused replay: no converged fused solve on the current matrix to replay");
    FusedSolve F{c};
    if (!fused_setup(c, F)) throw Error("fused replay: the fused iteration is not available");
    F.base = c.fused_replay.base;
    const int n = c.fused_replay.n;
    hipEvent_t e[3];
    for (auto& x : e) MS_CHECK(hipEventCreate(&x));
    for (int w = 0; w < 3; w++) {
        F.launch_S(n, nullptr, 1);
        F.launch_V(n + 1, false, 0, 0.0, 0.0, nullptr, 0, 1);
    }
    MS_CHECK(hipEventRecord(e[0], c.stream));
    for (int i = 0; i < n_launches; i++) F.launch_S(n, nullptr, 1);
    MS_CHECK(hipEventRecord(e[1], c.stream));
    for (int i = 0; i < n_launches; i++) F.launch_V(n + 1, false, 0, 0.0, 0.0, nullptr, 0, 1);
    MS_CHECK(hipEventRecord(e[2], c.stream));
    MS_CHECK(hipEventSynchronize(e[2]));
    float ms[2] = {0.f, 0.f};
    for (int i = 0; i < 2; i++) MS_CHECK(hipEventElapsedTime(&ms[i], e[i], e[i + 1]));
    for (auto& x : e) (void)hipEventDestroy(x);
    c.coll->check();
    c.fused_replay.valid = false;  // (V has moved the vectors on)
    if (s_us) *s_us = 1e3 * ms[0] / n_launches;
    if (v_us) *v_us = 1e3 * ms[1] / n_launches;
}

__global__ void k_copy_ctrl(const PcgCtrl* __restrict__ src, PcgCtrl* __restrict__ dst_host)
{
    if (threadIdx.x == 0) {
        *dst_host = *src;
        __threadfence_system();
    }
}
static double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// ---- option "cg_variant" = 1 on ONE GPU: the Chronopoulos-Gear iteration of the sharded solve without the windows -----------------------------
// Two launches per iteration instead of three: S (the solver's SpMV on u = M^-1 r, partial w.u) and V (k_cg_vec in its local mode: every
// workgroup re-reduces the partial sums, decides, updates p, s, x, r, u). Same iterates in exact arithmetic; p.Ap is delta - beta gamma /
// alpha_prev instead of a dot product of its own, one SpMV more per solve (w_0 = A u_0). NOT the default: the reference's loop
// (solve_pcg.h:170-225) is; kept as a measured alternative and as the one-GPU cross-check of the sharded iteration's arithmetic.
static void pcg_cg(Context& c, const double* rhs_dev, double abs_tol, double rel_tol, int max_iter, int stop_on_indef, mistark_pcg_info* info, double rhs_scale)
{
    const int gv = grid_for(c.nbr, BLOCK, PCG_GRID);
    BsrPart& m1 = c.part[1];
    const bool dyn = m1.nnzb > 0;
    double* part_wu = c.partials.p;
    double* pr[2][2] = {{c.partials.p + 4 * MAX_PARTIALS, c.partials.p + 5 * MAX_PARTIALS}, {c.partials.p + 2 * MAX_PARTIALS, c.partials.p + 3 * MAX_PARTIALS}};  // (r.u, r.r) by parity of k
    c.p2.ensure((size_t)c.ndofs);
    if (c.perm_active) c.xl.ensure((size_t)c.ndofs);
    double* const xs = c.perm_active ? c.xl.p : c.du.p;
    double *u = c.z.p, *w = c.q.p, *p = c.p.p, *s = c.p2.p, *r = c.r.p;
    {
        // prologue as in pcg(): preconditioner, x = 0, r = b, u = M^-1 r; partial (r.r, r.u) where V_1 expects those of "V_0" (parity 0)
        const BsrPart& d1 = c.part[1];
        hipLaunchKernelGGL(k_pcg_prologue, dim3(gv), dim3(BLOCK), 0, c.stream, rhs_dev, rhs_scale, (const float*)c.part[0].vals.p, (const int32_t*)c.diag_slot[0].p,
                           d1.nnzb ? (const float*)d1.vals.p : (const float*)nullptr, (const int32_t*)c.diag_slot[1].p, c.nbr, c.dinv.p, xs, r, u, p, pr[0][1], pr[0][0],
                           c.perm_active ? (const int32_t*)c.iperm.p : (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_pcg_init2, dim3(1), dim3(BLOCK), 0, c.stream, pr[0][1], pr[0][0], gv, abs_tol, c.ctrl.p, 1);
    }
    constexpr int BATCH = 4;
    PcgCtrl* hs[2] = {reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048), reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048 + 64)};
    const int epoch = ++c.pcg_epoch;
    int k = 1;
    bool tail_done = false;
    CgFast f{};
    auto launch_batch = [&](int slot) {
        hs[slot]->epoch = epoch - 1;
        hs[slot]->done = 0;
        hs[slot]->n_iter = -1;
        const int k_end = std::min(max_iter + 1, k + BATCH - 1);
        for (; k <= k_end; k++) {
            const bool check_only = k == max_iter + 1;
            const int gs = launch_spmv<0>(c, u, w, u, part_wu, c.ctrl.p, /*combine=*/false, nullptr);  // w_{k-1} = A u_{k-1}, partial w.u
            hipLaunchKernelGGL(k_cg_vec, dim3(gv), dim3(BLOCK), 0, c.stream, k, check_only ? 1 : 0, stop_on_indef, abs_tol, rel_tol, f, 0u, 0u, (const float*)c.dinv.p, c.nbr, u,
                               (const double*)w, p, s, xs, r, c.ctrl.p, dyn ? (const int32_t*)m1.crow_of_row.p : (const int32_t*)nullptr, (const uint32_t*)m1.row_chunk0.p,
                               (const double*)m1.yd.p, (const double*)m1.chunk_partial.p, (const int32_t*)nullptr, (const uint32_t*)nullptr, pr[k & 1][0], pr[k & 1][1],
                               k == k_end ? hs[slot] : (PcgCtrl*)nullptr, epoch, 0, (const double*)part_wu, gs, (const double*)pr[(k - 1) & 1][0], (const double*)pr[(k - 1) & 1][1], gv, 0);
            if (check_only) tail_done = true;
        }
        return k_end;
    };
    PcgCtrl h{};
    int slot = 0;
    int k_end_cur = launch_batch(0);
    for (;;) {
        const bool more = !tail_done;
        int k_end_next = 0;
        if (more) k_end_next = launch_batch(slot ^ 1);
        const volatile PcgCtrl* v = hs[slot];
        const double t_wait = now_seconds();
        auto reported = [&] { return v->epoch == epoch && (v->done || v->n_iter >= k_end_cur); };
        for (uint64_t spins = 0; !reported(); spins++) {
            __builtin_ia32_pause();
            if ((spins & 0xfffff) != 0xfffff) continue;
            const hipError_t q = hipStreamQuery(c.stream);
            if (q != hipErrorNotReady) {
                MS_CHECK(q);
                if (!reported()) {
                    PcgCtrl dev{};
                    MS_CHECK(hipMemcpy(&dev, c.ctrl.p, sizeof(PcgCtrl), hipMemcpyDeviceToHost));
                    hs[slot]->converged = dev.converged;
                    hs[slot]->indef = dev.indef;
                    hs[slot]->error = dev.error;
                    hs[slot]->n_iter = dev.done ? dev.n_iter : k_end_cur;
                    hs[slot]->done = dev.done ? 1 : 0;
                    hs[slot]->epoch = epoch;
                }
                break;
            }
            if (now_seconds() - t_wait > 60.0) throw Error("pcg: the device did not report iteration " + std::to_string(k_end_cur) + " within 60 s");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        h = *hs[slot];
        if (h.done || !more) break;
        slot ^= 1;
        k_end_cur = k_end_next;
    }
    if (c.perm_active) rows_from_solver(c, xs, c.du.p);
    const int n_it = h.done ? h.n_iter : max_iter;
    c.last_cg_iters = n_it;
    if (info) {
        info->converged = h.done ? h.converged : 0;
        info->n_iterations = n_it;
        info->found_indefiniteness = h.indef;
        info->error = h.error;
        info->reserved = 0;
    }
}
// SpMV timing inside the solver: every SPMV_SAMPLE-th launch is bracketed by a pair of pooled HIP events on the engine's stream
constexpr int SPMV_SAMPLE = 32;  // (a sampled launch costs the stream ~14 us of marker packets: 1.3 % of the timed region at every 16th launch, measured)
void pcg(Context& c, const double* rhs_dev, double abs_tol, double rel_tol, int max_iter, int stop_on_indef, mistark_pcg_info* info, double rhs_scale)
{
    if (!c.have_matrix) throw Error("pcg: matrix not assembled");
    if (c.world > 1) {
        if (rhs_scale != 1.0) throw Error("pcg: a scaled right-hand side is a single-GPU shortcut");
        if (pcg_sharded_fused(c, rhs_dev, abs_tol, rel_tol, max_iter, stop_on_indef, info)) {
            c.n_fused_solves++;
        } else {
            pcg_sharded(c, rhs_dev, abs_tol, rel_tol, max_iter, stop_on_indef, info);
            c.n_unfused_solves++;
        }
        return;
    }
    if (c.cg_variant == 1) {
        pcg_cg(c, rhs_dev, abs_tol, rel_tol, max_iter, stop_on_indef, info, rhs_scale);
        return;
    }
    const int gv = grid_for(c.nbr, BLOCK, PCG_GRID);  // one block row per thread up to 262 144 block rows
    BsrPart& m1 = c.part[1];
    const bool dyn = m1.nnzb > 0;
    double* part_pq = c.partials.p;
    double* part_rr = c.partials.p + MAX_PARTIALS;
    double* part_rz = c.partials.p + 2 * MAX_PARTIALS;
    double* part_bb = c.partials.p + 3 * MAX_PARTIALS;
    const bool fuse_dir = !c.no_fuse_dir;
    c.p2.ensure((size_t)c.ndofs);
    // (solver numbering: the solution accumulates in a scratch vector and is written to c.du in the caller's numbering at the end)
    if (c.perm_active) c.xl.ensure((size_t)c.ndofs);
    double* const xs = c.perm_active ? c.xl.p : c.du.p;
    // (fused: iteration 1 reads p_0 = buffer 0 with beta = 0; k_pcg_init leaves z there, so 0 * p_0 is finite)
    {
        const BsrPart& d1 = c.part[1];
        hipLaunchKernelGGL(k_pcg_prologue, dim3(gv), dim3(BLOCK), 0, c.stream, rhs_dev, rhs_scale, (const float*)c.part[0].vals.p, (const int32_t*)c.diag_slot[0].p,
                           d1.nnzb ? (const float*)d1.vals.p : (const float*)nullptr, (const int32_t*)c.diag_slot[1].p, c.nbr, c.dinv.p, xs, c.r.p, c.z.p, fuse_dir ? c.p2.p : c.p.p,
                           part_bb, part_rz, c.perm_active ? (const int32_t*)c.iperm.p : (const int32_t*)nullptr);
    }
    hipLaunchKernelGGL(k_pcg_init2, dim3(1), dim3(BLOCK), 0, c.stream, part_bb, part_rz, gv, abs_tol, c.ctrl.p, 1);
    // Iterations are launched in batches of PCG_BATCH; after each batch the control block is copied to a pinned slot and an
    // event recorded. The host launches batch b+1 BEFORE it waits for batch b's event, so the GPU never idles on the host's
    // convergence check, and at most one batch of device-side no-op launches (ctrl->done) is wasted after convergence.
    constexpr int PCG_BATCH_MAX = 8;  // (sizes of the sampling buffers)
    // (option "pcg_batch"; 0 = by size: 3 for the large systems, whose iterations are long enough for the host to keep up with shorter batches and
    // whose solves then queue fewer no-op launches behind the iteration that converged — configs[3]: 1.140 against 1.155 ms per solve, 2 / 3 / 4 / 6
    // = 1.145 / 1.140 / 1.155 / 1.176 —, 4 for the small ones, whose 13 us iterations the host barely outruns: configs[0] 254 against 244-248)
    const int PCG_BATCH = std::min(std::max(c.pcg_batch > 0 ? c.pcg_batch : (c.nbr >= 100000 ? 3 : 4), 1), PCG_BATCH_MAX);
    PcgCtrl* hs[2] = {reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048), reinterpret_cast<PcgCtrl*>(host_scratch(c, 4096) + 2048 + 64)};  // pinned
    while (c.pcg_ev.size() < 2) {
        hipEvent_t e;
        MS_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.pcg_ev.push_back(e);
    }
    const int epoch = ++c.pcg_epoch;
    std::vector<int> sampled[2];
    int clk_grid[2] = {0, 0};
    int k = 1;
    auto launch_batch = [&](int slot) {
        const int k_end = std::min(max_iter, k + PCG_BATCH - 1);
        // (the slot is written by the batch's last direction kernel; the host waits for it by watching the slot itself — an event record
        // between batches is a marker packet the next SpMV waits behind: 5 us per batch)
        hs[slot]->epoch = epoch - 1;  // (whatever a straggler of the previous solve writes here carries the previous epoch, too)
        hs[slot]->done = 0;
        hs[slot]->n_iter = -1;
        sampled[slot].clear();
        for (; k <= k_end; k++) {
            const bool sample = c.time_spmv && (k % SPMV_SAMPLE) == 0;
            const size_t e0 = (size_t)slot * 3 * PCG_BATCH_MAX + 3 * sampled[slot].size();
            if (sample) {
                while (c.ev.size() < (size_t)6 * PCG_BATCH_MAX) {
                    hipEvent_t e;
                    MS_CHECK(hipEventCreate(&e));
                    c.ev.push_back(e);
                }
                MS_CHECK(hipEventRecord(c.ev[e0], c.stream));
            }
            // p_k = z + beta p_{k-1} lives in buffer k & 1 (the SpMV forms it on the fly and stores it)
            double* pk = (k & 1) ? c.p.p : c.p2.p;
            const double* pprev = (k & 1) ? c.p2.p : c.p.p;
            int gs;
            if (fuse_dir) {
                gs = launch_spmv_dir(c, DirArgs{c.z.p, pprev, pk, part_rr, part_rz, gv, k, abs_tol, rel_tol}, c.q.p, part_pq);
            } else {
                pk = c.p.p;
                uint64_t* clk = nullptr;
                if (sample) {
                    if (!c.spmv_clk) MS_CHECK(hipHostMalloc((void**)&c.spmv_clk, sizeof(uint64_t) * 2 * PCG_BATCH_MAX * 2 * MAX_PARTIALS, hipHostMallocDefault));
                    clk = c.spmv_clk + ((size_t)slot * PCG_BATCH_MAX + sampled[slot].size()) * 2 * MAX_PARTIALS;
                    std::memset(clk, 0, sizeof(uint64_t) * 2 * MAX_PARTIALS);
                }
                gs = launch_spmv<0>(c, c.p.p, c.q.p, c.p.p, part_pq, c.ctrl.p, /*combine=*/false, clk);
                if (sample) clk_grid[slot] = gs;
            }
            if (sample) {
                MS_CHECK(hipEventRecord(c.ev[e0 + 1], c.stream));
                // an empty bracket right behind: what a pair of event records costs the stream by itself (the marker packets' own processing
                // is inside every bracketed duration; bench.py reports both figures)
                MS_CHECK(hipEventRecord(c.ev[e0 + 2], c.stream));
                sampled[slot].push_back(k);
            }
            hipLaunchKernelGGL(k_pcg_step, dim3(gv), dim3(BLOCK), 0, c.stream, k, stop_on_indef, part_pq, gs, c.dinv.p, c.nbr, (const double*)pk, c.q.p, xs, c.r.p, c.z.p, part_rr,
                               part_rz, c.ctrl.p, dyn ? (const int32_t*)m1.crow_of_row.p : nullptr, (const uint32_t*)m1.row_chunk0.p, (const double*)m1.yd.p,
                               (const double*)m1.chunk_partial.p);
            if (!fuse_dir)
                hipLaunchKernelGGL(k_pcg_dir, dim3(gv), dim3(BLOCK), 0, c.stream, k, abs_tol, rel_tol, part_rr, part_rz, gv, c.ndofs, c.z.p, c.p.p, c.ctrl.p, 1,
                                   k == k_end ? hs[slot] : (PcgCtrl*)nullptr, epoch);
        }
        // (fused: the test of the batch's last iteration would only run with the next batch's first SpMV; the host reads the control block now)
        if (fuse_dir) hipLaunchKernelGGL(k_pcg_check, dim3(1), dim3(BLOCK), 0, c.stream, DirArgs{c.z.p, nullptr, nullptr, part_rr, part_rz, gv, k_end + 1, abs_tol, rel_tol}, c.ctrl.p);
        // the control block reaches the pinned slot from the batch's last k_pcg_dir itself (round 1: a copy command on another engine, 4 us
        // + a 5.6 us gap; then a one-wavefront copy kernel, 4 us + its boundary, every four iterations); the fused variant still copies
        if (fuse_dir) {
            hipLaunchKernelGGL(k_copy_ctrl, dim3(1), dim3(64), 0, c.stream, (const PcgCtrl*)c.ctrl.p, hs[slot]);
            MS_CHECK(hipEventRecord(c.pcg_ev[slot], c.stream));
        }
        return k_end;
    };
    auto drain = [&](int slot, int last_real_iter) {
        for (size_t i = 0; i < sampled[slot].size(); i++) {
            if (sampled[slot][i] > last_real_iter) continue;  // early-exit launch after convergence
            float ms = 0.f;
            const size_t e0 = (size_t)slot * 3 * PCG_BATCH_MAX + 3 * i;
            float ms_empty = 0.f;
            if (hipEventElapsedTime(&ms, c.ev[e0], c.ev[e0 + 1]) == hipSuccess && hipEventElapsedTime(&ms_empty, c.ev[e0 + 1], c.ev[e0 + 2]) == hipSuccess) {
                c.spmv_ms_sum += ms;
                c.spmv_empty_ms_sum += ms_empty;
                c.spmv_n++;
            }
            if (c.spmv_clk && clk_grid[slot] > 0) {  // the same launch on the device clock
                const uint64_t* clk = c.spmv_clk + ((size_t)slot * PCG_BATCH_MAX + i) * 2 * MAX_PARTIALS;
                uint64_t t0 = ~0ull, t1 = 0;
                bool complete = true;
                for (int b = 0; b < clk_grid[slot]; b++) {
                    if (clk[2 * b] == 0 || clk[2 * b + 1] == 0) { complete = false; break; }
                    t0 = std::min(t0, clk[2 * b]);
                    t1 = std::max(t1, clk[2 * b + 1]);
                }
                if (complete && t1 > t0) {
                    c.spmv_clk_ticks += (double)(t1 - t0);
                    c.spmv_clk_n++;
                }
            }
        }
    };
    PcgCtrl* h = nullptr;
    int slot = 0;
    int k_end_cur = launch_batch(0);
    // The look-ahead batch is held back when the batch in flight is expected to converge: from the errors the last two finished batches
    // reported, error_b ~ error_{b-1} * (error_{b-1} / error_{b-2}). A converged solve then wastes the rest of ONE batch instead of that
    // plus a whole batch of no-op launches (4 to 7 iterations of three launches each were 4 % of a solve); a wrong guess costs one host
    // round trip with the GPU idle. Same iterations either way. MEASURED on configs[3] (tools/ab_option.sh pcg_holdback 3): 1.150 ms per solve
    // with it, 1.140 without — the no-op launches are 2 us each and the wrong guesses cost as much as the right ones save: option pcg_holdback,
    // off by default.
    const double tol = std::max(abs_tol, rel_tol);
    double err1 = 1.0, err2 = -1.0;  // batch-end errors, newest first (error_0 = 1)
    for (;;) {
        const bool more = k <= max_iter;
        int k_end_next = 0;
        bool hold = false;
        if (more && c.pcg_holdback && !fuse_dir) {
            const double shrink = err2 > 0.0 ? std::min(1.0, std::max(0.02, err1 / err2)) : 0.5;
            hold = err1 * shrink < tol;
        }
        if (more && !hold) k_end_next = launch_batch(slot ^ 1);  // keep the GPU fed while the host looks at the previous batch
        project_speculate_pending(c);  // (a projection round to run beside this solve: queued behind the solve's first batches)
        if (fuse_dir) {
            MS_CHECK(hipEventSynchronize(c.pcg_ev[slot]));
        } else {
            const volatile PcgCtrl* v = hs[slot];
            const double t_wait = now_seconds();
            auto reported = [&] { return v->epoch == epoch && (v->done || v->n_iter >= k_end_cur); };
            for (uint64_t spins = 0; !reported(); spins++) {
                __builtin_ia32_pause();
                if ((spins & 0x3f) == 0) project_spec_poll(c);  // (a projection round started ahead of this solve: its second phase once its counts are here)
                if ((spins & 0xfffff) != 0xfffff) continue;
                // now and then a real look at the stream, as publish() does: a failed launch surfaces as its error, and a stream that has
                // drained without the slot being written (host memory the device's writes do not reach while kernels run) is answered from
                // the device's own control block instead of a time-out
                const hipError_t q = hipStreamQuery(c.stream);
                if (q != hipErrorNotReady) {
                    MS_CHECK(q);
                    if (!reported()) {
                        PcgCtrl dev{};
                        MS_CHECK(hipMemcpy(&dev, c.ctrl.p, sizeof(PcgCtrl), hipMemcpyDeviceToHost));
                        hs[slot]->converged = dev.converged;
                        hs[slot]->indef = dev.indef;
                        hs[slot]->error = dev.error;
                        hs[slot]->n_iter = dev.done ? dev.n_iter : k_end_cur;
                        hs[slot]->done = dev.done ? 1 : 0;
                        hs[slot]->epoch = epoch;
                    }
                    break;
                }
                if (now_seconds() - t_wait > 60.0) throw Error("pcg: the device did not report batch " + std::to_string(k_end_cur) + " within 60 s");
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        h = hs[slot];
        if (c.time_spmv) drain(slot, h->done ? h->n_iter : k_end_cur);
        if (h->done || !more) break;
        err2 = err1;
        err1 = h->error;
        if (hold) k_end_next = launch_batch(slot ^ 1);  // (the guess fell short)
        slot ^= 1;
        k_end_cur = k_end_next;
    }
    // (a look-ahead batch launched after convergence consists of device-side no-ops; later work queues behind it on the same stream)
    if (c.perm_active) rows_from_solver(c, xs, c.du.p);
    const int n_it = h->done ? h->n_iter : max_iter;
    c.last_cg_iters = n_it;
    if (info) {
        info->converged = h->done ? h->converged : 0;
        info->n_iterations = n_it;
        info->found_indefiniteness = h->indef;
        info->error = h->error;
        info->reserved = 0;
    }
}

Context::~Context()
{
    if (std::getenv("MISTARK_PRELAUNCH_STATS")) std::fprintf(stderr, "mistark: rank %d of %d: evaluation kernels started ahead: %lld taken over, %lld dropped\n", rank, world, (long long)n_prelaunch_used, (long long)n_prelaunch_dropped);
    contact_destroy(contact);
    direct_mf_destroy(llt_mf);
    if (dry) return;
    if (pre_stream) {
        (void)hipStreamSynchronize(pre_stream);
        (void)hipStreamDestroy(pre_stream);
    }
    for (EvalPre& q : pre)
        if (q.ev_in) {
            (void)hipEventDestroy(q.ev_in);
            (void)hipEventDestroy(q.ev_out);
        }
    for (int k = 0; k < 2; k++) {
        if (h_stage[k]) (void)hipHostFree(h_stage[k]);
        if (h_stage_ev[k]) (void)hipEventDestroy(h_stage_ev[k]);
    }
    for (auto e : ev) (void)hipEventDestroy(e);
    for (auto e : pcg_ev) (void)hipEventDestroy(e);
    for (auto e : stage_ev) (void)hipEventDestroy(e);
    if (h_scratch) (void)hipHostFree(h_scratch);
    if (h_pin) (void)hipHostFree(h_pin);
    if (pub) (void)hipHostFree(pub);
    if (pub2) (void)hipHostFree(pub2);
    if (spmv_clk) (void)hipHostFree(spmv_clk);
    if (spmv_clk_sharded) (void)hipHostFree(spmv_clk_sharded);
    if (aux_stream) {
        (void)hipStreamDestroy(aux_stream);
        for (auto& e : aux_ev)
            if (e) (void)hipEventDestroy(e);
    }
    if (side_stream) {
        (void)hipStreamDestroy(side_stream);
        (void)hipEventDestroy(side_ev[0]);
        (void)hipEventDestroy(side_ev[1]);
    }
    if (spec.stream) {  // (= pre_stream, destroyed above)
        (void)hipEventDestroy(spec.ev_in);
        (void)hipEventDestroy(spec.ev_done);
        (void)hipHostFree(spec.pinned);
    }
    if (stream && owns_stream) (void)hipStreamDestroy(stream);
}

}  // namespace mistark
