// Microbenchmark: cost of a device-wide barrier inside a persistent kernel on MI355X (all workgroups co-resident), with the
// release/acquire fences a producer/consumer exchange through global memory needs. usage: grid_barrier [wgs] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE);   // agent scope by default for __atomic builtins on global memory
        while (__atomic_load_n(bar, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* bar, double* buf, int iters, double* out)
{
    const unsigned G = gridDim.x;
    double acc = 0.0;
    const int me = blockIdx.x * 256 + threadIdx.x;
    const int n = G * 256;
    for (int it = 1; it <= iters; it++) {
        if (MODE >= 1) buf[me] = (double)it + me;                 // produce
        if (MODE >= 1) __threadfence();
        grid_barrier(bar, G * (unsigned)it);
        if (MODE >= 1) {
            __threadfence();
            const int other = (me + 256 * 37 + 11) % n;               // consume another workgroup's value
            const double v = __builtin_nontemporal_load(&buf[other]);
            acc += v - ((double)it + other);
            grid_barrier(bar + 32, G * (unsigned)it);              // second barrier so that nobody overwrites before all have read
        }
    }
    if (MODE >= 1) out[me] = acc;
}
int main(int argc, char** argv)
{
    int wgs = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned* bar; double *buf, *out;
    CK(hipMalloc(&bar, 4096)); CK(hipMalloc(&buf, (size_t)wgs * 256 * 8)); CK(hipMalloc(&out, (size_t)wgs * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(bar, 0, 4096));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(wgs), dim3(256), 0, 0, bar, buf, iters, out);
            else hipLaunchKernelGGL(k_bar<1>, dim3(wgs), dim3(256), 0, 0, bar, buf, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                double bad = 0;
                if (mode == 1) {
                    double* h = (double*)malloc((size_t)wgs * 256 * 8);
                    CK(hipMemcpy(h, out, (size_t)wgs * 256 * 8, hipMemcpyDeviceToHost));
                    for (int i = 0; i < wgs * 256; i++) bad += h[i] != 0.0;
                    free(h);
                }
                printf("wgs %d mode %d: %.2f us per iteration (%d barrier%s/iteration), stale reads %g\n", wgs, mode, 1e3 * ms / iters, mode + 1, mode ? "s" : "", bad);
            }
        }
    }
    return 0;
}
