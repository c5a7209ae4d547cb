// kernels_common.hpp — what the translation units of the engine's kernels share (kernels.hip: registry, element evaluation, reductions, prepare / pattern,
// eval; project.hip: PSD projection; solve.hip: assembly, SpMV, PCG): launch constants, wavefront / workgroup reductions, and the host helpers one unit
// defines and another uses.
#pragma once
#include <atomic>
#include <chrono>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "dist.hpp"
#include "ipc_dev.hpp"
#include "engine.hpp"

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

constexpr uint32_t NO_SRC = 0xFFFFFFFFu;
constexpr int CHUNK_BLOCKS = 256;
constexpr int DYN_SHORT_ROW = 32;  // contact rows of a node hold a handful of blocks; only the rows of rigid bodies in contact are long
constexpr uint32_t LONG_SLOT = 48;  // BSR blocks with more contributions than this are summed by a whole wavefront (k_assemble_long)
constexpr uint32_t VERY_LONG_SLOT = 4096;  // ... and beyond this by VLONG_SPLIT wavefronts and a second pass (the blocks of a rigid body under 10^4..10^5 contacts)
constexpr int VLONG_SPLIT = 64;
__host__ __device__ constexpr int tet_pair_index(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }  // a <= b: 0..9 (float pool of the lazy tets)
// position of component `comp` (row-major 3x3) of BSR block `slot` inside the 64-block tile layout (see SpMV)
__device__ __forceinline__ size_t tile_val_index(uint32_t slot, int comp)
{
    const size_t base = (size_t)(slot >> 6) * 576;
    const uint32_t lane = slot & 63u;
    if (comp < 4) return base + lane * 4 + comp;
    if (comp < 8) return base + 256 + lane * 4 + (comp - 4);
    return base + 512 + lane;
}
// kernels.hip
double* host_scratch(Context& c, size_t n);
void launch_tet_closed_list(Context& c, Potential& P, const uint32_t* list, int n_list, double* H, int n_pool);
// solve.hip
void gather_part(Context& c, int part, const uint8_t* only_dirty);
void assemble_part(Context& c, int part);

}  // namespace mistark
