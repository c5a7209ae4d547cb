// Microbenchmark 3 (round 6): a barrier among W workgroups that all sit on ONE XCD (launched as 8 W workgroups of which those
// with blockIdx % 8 == 0 take part: workgroup ids go round robin over the 8 dies), with a producer / consumer exchange through that die's L2.
// Release = workgroup scope (the vL1D is write-through: a store that has been acknowledged is in the L2 all participants share; NO buffer_wbl2,
// which is what made the device-wide barrier of grid_barrier2 cost a whole-L2 write-back per arrival), acquire = agent scope (buffer_inv sc1:
// the reader's vL1D may hold the line from the previous round). Every spin is bounded: a broken assumption ends the kernel with an error word
// instead of hanging the device. usage: grid_barrier3 [W] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr unsigned SPIN_MAX = 4000000u;
__device__ __forceinline__ bool xcd_barrier(unsigned* bar, unsigned n_part, unsigned epoch, unsigned* err)
{
    __shared__ int failed;
    __syncthreads();
    if (threadIdx.x == 0) {
        failed = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_part * epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_MAX) {
                failed = 1;
                atomicExch(err, epoch);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return failed == 0;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* bar, double* buf, int iters, double* out, unsigned* err, unsigned* xcc_seen)
{
    if ((blockIdx.x & 7) != 0) return;
    const unsigned W = gridDim.x / 8, w = blockIdx.x / 8;
    if (threadIdx.x == 0) {  // which die this workgroup really runs on (HW_REG_XCC_ID, bits 3:0)
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_seen[w] = id & 0xf;
    }
    const int me = w * 256 + threadIdx.x, n = W * 256;
    double acc = 0.0;
    unsigned ep = 0;
    for (int it = 1; it <= iters; it++) {
        if (MODE >= 1) buf[me] = (double)it + me;
        if (!xcd_barrier(bar, W, ++ep, err)) return;
        if (MODE >= 1) {
            const int other = (me + 256 * 5 + 11) % n;
            acc += buf[other] - ((double)it + other);
            if (!xcd_barrier(bar, W, ++ep, err)) return;
        }
    }
    if (MODE >= 1) out[me] = acc;
}
int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *bar, *err, *xcc;
    double *buf, *out;
    CK(hipMalloc(&bar, 256)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&xcc, 4 * W));
    CK(hipMalloc(&buf, (size_t)W * 256 * 8)); CK(hipMalloc(&out, (size_t)W * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(bar, 0, 256)); CK(hipMemset(err, 0, 4)); CK(hipMemset(out, 0, (size_t)W * 256 * 8));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(8 * W), dim3(256), 0, 0, bar, buf, iters, out, err, xcc);
            else hipLaunchKernelGGL(k_bar<1>, dim3(8 * W), dim3(256), 0, 0, bar, buf, iters, out, err, xcc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                unsigned* hx = (unsigned*)malloc(4 * W); CK(hipMemcpy(hx, xcc, 4 * W, hipMemcpyDeviceToHost));
                int distinct = 0; bool seen[16] = {false};
                for (int i = 0; i < W; i++) if (!seen[hx[i] & 15]) { seen[hx[i] & 15] = true; distinct++; }
                double bad = 0;
                if (mode == 1) {
                    double* h = (double*)malloc((size_t)W * 256 * 8);
                    CK(hipMemcpy(h, out, (size_t)W * 256 * 8, hipMemcpyDeviceToHost));
                    for (int i = 0; i < W * 256; i++) bad += h[i] != 0.0;
                    free(h);
                }
                printf("W %d on one die (distinct XCC ids seen: %d, first %u) mode %d: %.3f us per iteration (%d barrier%s/iteration) = %.3f us per barrier, stale reads %g, timeout at epoch %u\n", W, distinct,
                       hx[0], mode, 1e3 * ms / iters, mode ? 2 : 1, mode ? "s" : "", 1e3 * ms / iters / (mode ? 2 : 1), bad, herr);
                free(hx);
            }
        }
    return 0;
}
