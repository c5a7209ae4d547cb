// Microbenchmark 2: two-level device-wide barrier (groups of workgroups arrive on their own counter, the last of a group on the root, the last
// of all releases per-group flags), fences by one thread per workgroup. usage: grid_barrier2 [wgs] [iters] [group]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int LINE = 32;  // unsigned per 128-byte line
__device__ __forceinline__ void tree_barrier(unsigned* bar, int group, int ngroups, unsigned epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x / group;
        const int gsize = min(group, (int)gridDim.x - g * group);
        unsigned* gc = bar + (1 + g) * LINE;            // group counter
        unsigned* rel = bar + (1 + ngroups + g) * LINE;  // group release flag
        __atomic_thread_fence(__ATOMIC_RELEASE);
        if (group >= (int)gridDim.x) {  // flat
            __atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED);
            while (__atomic_load_n(bar, __ATOMIC_RELAXED) < gridDim.x * epoch) __builtin_amdgcn_s_sleep(1);
        } else {
            if (__atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED) == (unsigned)gsize * epoch - 1) {
                if (__atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED) == (unsigned)ngroups * epoch - 1)
                    for (int k = 0; k < ngroups; k++) __atomic_store_n(bar + (1 + ngroups + k) * LINE, epoch, __ATOMIC_RELAXED);
            }
            while (__atomic_load_n(rel, __ATOMIC_RELAXED) < epoch) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* bar, double* buf, int iters, double* out, int group, int ngroups)
{
    const unsigned G = gridDim.x;
    double acc = 0.0;
    const int me = blockIdx.x * 256 + threadIdx.x;
    const int n = G * 256;
    unsigned ep = 0;
    for (int it = 1; it <= iters; it++) {
        if (MODE >= 1) buf[me] = (double)it + me;
        tree_barrier(bar, group, ngroups, ++ep);
        if (MODE >= 1) {
            const int other = (me + 256 * 37 + 11) % n;
            const double v = MODE == 1 ? __builtin_nontemporal_load(&buf[other]) : buf[other];
            acc += v - ((double)it + other);
            tree_barrier(bar, group, ngroups, ++ep);
        }
    }
    if (MODE >= 1) out[me] = acc;
}
int main(int argc, char** argv)
{
    int wgs = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 2000, group = argc > 3 ? atoi(argv[3]) : 16;
    int ngroups = (wgs + group - 1) / group;
    unsigned* bar; double *buf, *out;
    size_t bb = (size_t)(2 + 2 * ngroups) * LINE * 4;
    CK(hipMalloc(&bar, bb)); CK(hipMalloc(&buf, (size_t)wgs * 256 * 8)); CK(hipMalloc(&out, (size_t)wgs * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(bar, 0, bb));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(wgs), dim3(256), 0, 0, bar, buf, iters, out, group, ngroups);
            else if (mode == 1) hipLaunchKernelGGL(k_bar<1>, dim3(wgs), dim3(256), 0, 0, bar, buf, iters, out, group, ngroups);
            else hipLaunchKernelGGL(k_bar<2>, dim3(wgs), dim3(256), 0, 0, bar, buf, iters, out, group, ngroups);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                double bad = 0;
                if (mode >= 1) {
                    double* h = (double*)malloc((size_t)wgs * 256 * 8);
                    CK(hipMemcpy(h, out, (size_t)wgs * 256 * 8, hipMemcpyDeviceToHost));
                    for (int i = 0; i < wgs * 256; i++) bad += h[i] != 0.0;
                    free(h);
                }
                printf("wgs %d group %d mode %d: %.2f us per iteration (%d barrier%s/iteration), stale reads %g\n", wgs, group, mode, 1e3 * ms / iters, mode ? 2 : 1, mode ? "s" : "", bad);
            }
        }
    }
    return 0;
}
