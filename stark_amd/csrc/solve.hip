// solve.hip — assembly of the element blocks into the float BSR tiles, block-Jacobi inverse, SpMV and the PCG loops (single GPU, row-sharded, fused
// over IPC windows): BlockedSparseMatrix.h, solve_pcg.h:83-232, ElementHessians.cpp:224-256 of the reference.
#include "kernels_common.hpp"

namespace mistark {

// Where the gather assembly reads a contribution from: one descriptor per sorted key. Bit 31: float pool (elemHf) instead of the double pool
// (elemH), bit 30: read the stored block transposed (lazy potentials keep the upper block triangle only), bits 0..29: 3x3 block index in that
// pool. NO_SRC: no data (structural diagonal keys; multi-GPU: elements of other ranks, the sum over ranks restores them).
constexpr uint32_t DESC_FLOAT = 0x80000000u, DESC_TRANS = 0x40000000u, DESC_MASK = 0x3fffffffu;
constexpr uint32_t SYM_NONE = 0xFFFFFFFFu, SYM_SKIP = 0xFFFFFFFEu;  // BsrPart::sym
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
// a summed block into the tile layout at storage position pos, and — mirror != SYM_NONE — its transpose at the transposed block's position
__device__ __forceinline__ void store_block(float* __restrict__ vals, uint32_t pos, uint32_t mirror, const double* acc)
{
    float* tile = vals + (size_t)(pos >> 6) * 576;
    const uint32_t lane = pos & 63u;
    reinterpret_cast<float4*>(tile)[lane] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
    reinterpret_cast<float4*>(tile)[64 + lane] = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
    tile[512 + lane] = (float)acc[8];
    if (mirror != SYM_NONE) {
        float* t2 = vals + (size_t)(mirror >> 6) * 576;
        const uint32_t l2 = mirror & 63u;
        reinterpret_cast<float4*>(t2)[l2] = make_float4((float)acc[0], (float)acc[3], (float)acc[6], (float)acc[1]);
        reinterpret_cast<float4*>(t2)[64 + l2] = make_float4((float)acc[4], (float)acc[7], (float)acc[2], (float)acc[5]);
        t2[512 + l2] = (float)acc[8];
    }
}
__global__ __launch_bounds__(BLOCK) void k_assemble_gather(const double* __restrict__ elemH, const float* __restrict__ elemHf, const uint32_t* __restrict__ slot_start,
                                                           const uint32_t* __restrict__ sorted_src, int64_t nnzb, const uint32_t* __restrict__ store_slot, float* __restrict__ vals,
                                                           const uint8_t* __restrict__ only_dirty, const uint32_t* __restrict__ sym)
{
    const int64_t slot = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= nnzb) return;
    if (only_dirty && !only_dirty[store_slot ? store_slot[slot] : (uint32_t)slot]) return;  // (project(): only the blocks a projection round touched; flags by storage position)
    // (sym: block (i, j) of the static part whose contributions all come from the tets' float pool is the transpose of block (j, i) — the same pool
    // blocks, read the other way round: the lane of the upper one writes both, the lower one's lane leaves at once; see k_sym_classify)
    const uint32_t mirror = sym ? sym[slot] : SYM_NONE;
    if (mirror == SYM_SKIP) return;
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
    store_block(vals, pos, mirror, acc);
}
// ---- the same sums with the wavefront's long lists dealt out (round 5) ----------------------------------------------------------------------
// k_assemble_gather is bound by instruction issue, not by memory: a wavefront of 64 consecutive blocks holds about four diagonal blocks whose
// lists are 24 contributions long (every tet around the node) beside sixty lists of 4-6, so it runs six rounds of the four-way loop — ~500
// instructions each with the conversions to double — although 1.6 would do for the average lane (ablation on the 1M-tet matrix: without pool loads
// AND without stores the launch still takes 204 of 320 us; halving the pool reads through the mirror table gains 4 %, eight loads in flight per lane
// nothing). Here a lane sums its own list only when it is short (<= SPLIT_LEN: two rounds); the longer lists of the wavefront are then
// taken eight at a time by groups of eight lanes: lane s of a group adds the contributions s, s + 8, ... of the group's list, the eight partial sums
// are folded by three xor-shuffles (every lane adds the same pair of numbers, so all eight end with the same bits) and lane 0 of the group stores the
// block. A fixed order of additions per block as before — a DIFFERENT one for the long lists, so those sums may differ from k_assemble_gather's in
// the last bit of the double before it is rounded to float.
constexpr uint32_t SPLIT_LEN = 8;
__device__ __forceinline__ void add_contribution(uint32_t d, const double* __restrict__ elemH, const float* __restrict__ elemHf, double* acc)
{
    if (d == NO_SRC) return;
    if (d & DESC_FLOAT) {
        const F3* src = reinterpret_cast<const F3*>(elemHf + (size_t)(d & DESC_MASK) * 9);
        const F3 a = src[0], b = src[1], c = src[2];
        const bool t = (d & DESC_TRANS) != 0u;
        acc[0] += (double)a.x;
        acc[1] += (double)(t ? b.x : a.y);
        acc[2] += (double)(t ? c.x : a.z);
        acc[3] += (double)(t ? a.y : b.x);
        acc[4] += (double)b.y;
        acc[5] += (double)(t ? c.y : b.z);
        acc[6] += (double)(t ? a.z : c.x);
        acc[7] += (double)(t ? b.z : c.y);
        acc[8] += (double)c.z;
    } else {
        const double* h = elemH + (size_t)d * 9;
#pragma unroll
        for (int c = 0; c < 9; c++) acc[c] += h[c];
    }
}
__global__ __launch_bounds__(BLOCK) void k_assemble_gather_split(const double* __restrict__ elemH, const float* __restrict__ elemHf, const uint32_t* __restrict__ slot_start,
                                                                 const uint32_t* __restrict__ sorted_src, int64_t nnzb, const uint32_t* __restrict__ store_slot,
                                                                 float* __restrict__ vals, const uint32_t* __restrict__ sym)
{
    // (no lane leaves before the cooperative part: the shuffles below need the whole wavefront)
    const int64_t slot = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t k0 = 0, k1 = 0, mirror = SYM_NONE;
    bool active = slot < nnzb;
    if (active) {
        mirror = sym ? sym[slot] : SYM_NONE;
        active = mirror != SYM_SKIP;
    }
    if (active) {
        k0 = slot_start[slot];
        k1 = slot_start[slot + 1];
        active = k1 - k0 <= LONG_SLOT;  // (beyond: k_assemble_long / the very long lists' two passes)
    }
    const uint32_t len = active ? k1 - k0 : 0u;
    const bool is_long = len > SPLIT_LEN;
    if (active && !is_long) {  // own short list: the four-way loop of k_assemble_gather
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t kb = k0; kb < k1; kb += 4) {
            uint32_t d[4];
            F3 v[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = kb + u < k1 ? sorted_src[kb + u] : NO_SRC;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool f = d[u] != NO_SRC && (d[u] & DESC_FLOAT) != 0u;
                const F3* src = reinterpret_cast<const F3*>(elemHf + (f ? (size_t)(d[u] & DESC_MASK) * 9 : (size_t)0));
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
            if (any_double) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (d[u] == NO_SRC || (d[u] & DESC_FLOAT)) continue;
                    const double* h = elemH + (size_t)d[u] * 9;
#pragma unroll
                    for (int c = 0; c < 9; c++) acc[c] += h[c];
                }
            }
        }
        store_block(vals, store_slot ? store_slot[slot] : (uint32_t)slot, mirror, acc);
    }
    // the wavefront's long lists, eight at a time
    unsigned long long todo = __ballot(is_long);
    const int sub = lane & 7, grp = lane >> 3;
    while (todo != 0ull) {  // (wave-uniform)
        int src = -1;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (todo == 0ull) break;
            const int bit = __ffsll((long long)todo) - 1;
            if (g == grp) src = bit;
            todo &= todo - 1ull;
        }
        const bool has = src >= 0;
        const int from = has ? src : lane;
        const uint32_t g_k0 = (uint32_t)__shfl((int)k0, from, 64), g_k1 = (uint32_t)__shfl((int)k1, from, 64), g_mirror = (uint32_t)__shfl((int)mirror, from, 64);
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (has)
            for (uint32_t k = g_k0 + (uint32_t)sub; k < g_k1; k += 8u) add_contribution(sorted_src[k], elemH, elemHf, a);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
            for (int c = 0; c < 9; c++) a[c] += __shfl_xor(a[c], o, 64);
        }
        if (has && sub == 0) {
            const int64_t gs = slot - lane + src;
            store_block(vals, store_slot ? store_slot[gs] : (uint32_t)gs, g_mirror, a);
        }
    }
}
// Which blocks of the static part are written by their transpose's lane (k_assemble_gather, `sym`). Block (i, j), i != j, qualifies when every
// contribution of it AND of block (j, i) is a block of the lazy tets' float pool: that pool holds the upper-triangle block pairs of an element once,
// so the two lists name the same pool blocks with opposite transposition flags (checked: equal length, equal sum and xor of the pool indices, the
// flags add up) and the two sums are transposes of each other up to the order of the additions (double accumulators, one rounding to float at the
// end). The lane of the upper block (i < j) then stores both; the gather reads 10 instead of 16 pool blocks per tet. Everything else — diagonal
// blocks, blocks with a contribution from the double pool, long lists — keeps its own lane (SYM_NONE).
__global__ __launch_bounds__(BLOCK) void k_sym_classify(const uint32_t* __restrict__ slot_start, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ colw,
                                                        const uint32_t* __restrict__ slot_row, const int64_t* __restrict__ row_ptr, int64_t nnzb, const uint32_t* __restrict__ store_slot,
                                                        uint32_t* __restrict__ sym)
{
    const int64_t slot = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= nnzb) return;
    uint32_t out = SYM_NONE;
    const uint32_t r = slot_row[slot], c = colw[slot] & 0x7fffffffu;
    const uint32_t k0 = slot_start[slot], k1 = slot_start[slot + 1], len = k1 - k0;
    if (r != c && len > 0 && len <= LONG_SLOT) {
        int64_t lo = row_ptr[c], hi = row_ptr[c + 1];  // (the static part holds every block row: compact row = row)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((colw[mid] & 0x7fffffffu) < r) lo = mid + 1;
            else hi = mid;
        }
        const int64_t t = lo;
        if (t < row_ptr[c + 1] && (colw[t] & 0x7fffffffu) == r && slot_row[t] == c) {
            const uint32_t t0 = slot_start[t], t1 = slot_start[t + 1];
            if (t1 - t0 == len) {
                bool pure = true;
                uint32_t sum_a = 0, sum_b = 0, xor_a = 0, xor_b = 0, n_trans = 0;
                for (uint32_t k = 0; k < len; k++) {
                    const uint32_t a = desc[k0 + k], b = desc[t0 + k];
                    pure = pure && a != NO_SRC && b != NO_SRC && (a & DESC_FLOAT) && (b & DESC_FLOAT);
                    sum_a += a & DESC_MASK; sum_b += b & DESC_MASK;
                    xor_a ^= a & DESC_MASK; xor_b ^= b & DESC_MASK;
                    n_trans += ((a & DESC_TRANS) ? 1u : 0u) + ((b & DESC_TRANS) ? 1u : 0u);
                }
                if (pure && sum_a == sum_b && xor_a == xor_b && n_trans == len) out = r < c ? (store_slot ? store_slot[t] : (uint32_t)t) : SYM_SKIP;
            }
        }
    }
    sym[slot] = out;
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
    m.sym_valid = false;
}
// the blocks of a matrix part summed from the pools in sorted-key order; only_dirty: just the flagged blocks (BsrPart::slot_dirty)
void gather_part(Context& c, int part, const uint8_t* only_dirty)
{
    BsrPart& m = c.part[part];
    make_descriptors(c, part);
    const uint32_t* store = part == 0 ? m.store_slot.p : nullptr;
    const uint32_t* desc = m.sorted_desc.p;
    // the whole static part from the lazy float pool, one GPU: blocks that are each other's transposes are summed once (k_sym_classify)
    const uint32_t* sym = nullptr;
    if (part == 0 && !only_dirty && c.world == 1 && c.lazy_active && !c.no_sym_gather && !c.hf_layout && m.n_keys > 0) {
        if (!m.sym_valid) {
            m.sym.ensure((size_t)m.nnzb);
            hipLaunchKernelGGL(k_sym_classify, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, m.slot_start.p, desc, m.colw.p, m.slot_row.p, m.row_ptr.p, m.nnzb, store, m.sym.p);
            m.sym_valid = true;
        }
        sym = m.sym.p;
    }
    // a whole part: the wavefront's long lists dealt out (k_assemble_gather_split); the blocks a projection round touched: one lane per block
    if (!only_dirty && !c.no_split_gather)
        hipLaunchKernelGGL(k_assemble_gather_split, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.nnzb, store, m.vals.p, sym);
    else
        hipLaunchKernelGGL(k_assemble_gather, dim3(grid_for(m.nnzb)), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.nnzb, store, m.vals.p, only_dirty, sym);
    if (m.n_long > 0)
        hipLaunchKernelGGL(k_assemble_long, dim3((m.n_long + 3) / 4), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.long_slots.p, m.n_long, store, m.vals.p, only_dirty);
    if (m.n_vlong > 0) {
        c.vlong_part.ensure((size_t)m.n_vlong * VLONG_SPLIT * 9);
        hipLaunchKernelGGL(k_assemble_vlong_part, dim3((m.n_vlong * VLONG_SPLIT + 3) / 4), dim3(BLOCK), 0, c.stream, c.elemH.p, c.elemHf.p, m.slot_start.p, desc, m.vlong_slots.p, m.n_vlong,
                           c.vlong_part.p, only_dirty, store);
        hipLaunchKernelGGL(k_assemble_vlong_fold, dim3((m.n_vlong + 3) / 4), dim3(BLOCK), 0, c.stream, (const double*)c.vlong_part.p, m.vlong_slots.p, m.n_vlong, store, m.vals.p, only_dirty);
    }
}
void assemble_part(Context& c, int part)
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
    for (int k = 0; k < 4; k++) {
        if (h_small[k]) (void)hipHostFree(h_small[k]);
        if (h_small_ev[k]) (void)hipEventDestroy(h_small_ev[k]);
    }
    for (auto& kv : pinned)  // (option pin_host_arrays: the caller's arrays are the caller's again)
        if (kv.second.ok) (void)hipHostUnregister(const_cast<void*>(kv.first));
    (void)hipGetLastError();
    for (hipEvent_t e : evt)  // (MISTARK_EVAL_EVENTS marks, kernels.hip evt_mark)
        if (e) (void)hipEventDestroy(e);
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
