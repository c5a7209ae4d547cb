// potargs.hpp — the kernel argument block of one potential. Plain data, no includes beyond fixed-width integers: the same text is compiled into
// libmistark.so and handed to hipRTC in front of the kernels custom.hip emits for user-defined potentials (csrc/Makefile: rtc_src.inc).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstdint>
#else
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#endif

namespace mistark {

constexpr int MAX_BIND = 40;
constexpr int MAX_NB = 8;

// Kernel argument block of one potential (device pointers)
struct PotArgs
{
    const double* arr[MAX_BIND];
    int conn_col[MAX_BIND];
    const int32_t* conn;
    int conn_stride;
    int n_elem;
    int e_begin, e_count;     // this rank's contiguous element range (multi-GPU sharding; the whole table on one GPU)
    const uint32_t* elem_list;  // != nullptr: the kernel's element le is elem_list[le] (le < e_count) and pools are indexed by le
    int n_pool;               // elements per block pair in the element-Hessian pool (pool stride)
    double* gpool;            // != nullptr: gradient contributions go to gpool[(k * n_gpool + pool position) * 3 + i] (summed by k_grad_gather) instead of atomics
    int n_gpool;
    // sharded runs (shard.hip): local index of a global block row ([0, n_own): owned, then ghosts, -1: neither); nullptr on one GPU.
    // An element's energy counts on the rank that owns the row of its first DoF block.
    const int32_t* lrow;
    int n_own;
    int dbg;                  // measurement switches (option "kernel_dbg"; results are wrong when set)
    int dof_col[MAX_NB];      // connectivity column providing the node of local DoF block k
    int dof_row_off[MAX_NB];  // first block row of the DoF set of local DoF block k
    // Gradient rows of SMALL DoF sets (a handful of rigid bodies touched by tens of thousands of contacts) are not accumulated in place:
    // 68 k atomics on the same six addresses serialise (1.6 ms per contact kind on configs[2]). Their contributions go to one of
    // HOT_WAYS copies chosen by the workgroup index and are folded into the gradient after the last potential (k_fold_hot).
    int hot_base[MAX_NB];     // index of the set's first row among the hot rows, -1: accumulate in place
    double* grad_hot;         // [HOT_WAYS][n_hot][3]
    int n_hot;
};
constexpr int HOT_WAYS = 64;

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
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

#endif

}  // namespace mistark
