// Micro-benchmark (round 5): a chain of small dependent launches like the contact search's box build — count, exclusive scan (hipcub), fill,
// radix sort of 41-bit keys with payload (hipcub), gather — issued launch by launch, against the same chain captured once into a hipGraph and
// replayed. Question: is the chain host-launch-bound, and does the library survive stream capture on this ROCm?
// build: hipcc --offload-arch=gfx950 -O3 graph_chain.hip -o graph_chain ; run on the GPU box
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_count(const float* x, int n, uint32_t* cnt) { int i = blockIdx.x * 256 + threadIdx.x; if (i <= n) cnt[i] = i < n ? 1u + (uint32_t)(x[i] > 0.5f) : 0u; }
__global__ void k_fill(const float* x, int n, const uint32_t* off, uint64_t* keys, uint32_t* idx) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    for (uint32_t k = off[i]; k < off[i + 1]; k++) { keys[k] = ((uint64_t)(i % 192) << 32) | __float_as_uint(x[i]); idx[k] = i; } }
__global__ void k_gather(const uint64_t* keys, const uint32_t* idx, int n, float* out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = (float)(keys[i] & 0xffff) + idx[i]; }
int main() {
    const int n = 70000, cap = 2 * n;
    float *x, *out; uint32_t *cnt, *off, *idx, *idx2; uint64_t *keys, *keys2; void* tmp; size_t tb = 1 << 24;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, cap * 4)); CK(hipMalloc(&cnt, (n + 1) * 4)); CK(hipMalloc(&off, (n + 1) * 4));
    CK(hipMalloc(&idx, cap * 4)); CK(hipMalloc(&idx2, cap * 4)); CK(hipMalloc(&keys, cap * 8)); CK(hipMalloc(&keys2, cap * 8)); CK(hipMalloc(&tmp, tb));
    std::vector<float> h(n); for (int i = 0; i < n; i++) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f;
    CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto chain = [&]() -> hipError_t {
        hipMemsetAsync(keys, 0xff, cap * 8, s);
        hipLaunchKernelGGL(k_count, dim3((n + 256) / 256), dim3(256), 0, s, x, n, cnt);
        size_t t = tb; hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, t, cnt, off, n + 1, s); if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, s, x, n, off, keys, idx);
        hipcub::DoubleBuffer<uint64_t> dk(keys, keys2); hipcub::DoubleBuffer<uint32_t> dv(idx, idx2);
        t = tb; e = hipcub::DeviceRadixSort::SortPairs(tmp, t, dk, dv, cap, 0, 41, s); if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gather, dim3((cap + 255) / 256), dim3(256), 0, s, dk.Current(), dv.Current(), cap, out);
        return hipGetLastError(); };
    for (int w = 0; w < 5; w++) CK(chain());
    CK(hipStreamSynchronize(s));
    const int reps = 200;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) CK(chain());
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    std::printf("stream launches: host issue %.1f us per chain, wall %.1f us per chain\n", 1e6 * std::chrono::duration<double>(t1 - t0).count() / reps, 1e6 * std::chrono::duration<double>(t2 - t0).count() / reps);
    // one chain + sync each time (what a search that ends in a read-back sees)
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) { CK(chain()); CK(hipStreamSynchronize(s)); }
    t2 = std::chrono::steady_clock::now();
    std::printf("stream launches + sync each: wall %.1f us per chain\n", 1e6 * std::chrono::duration<double>(t2 - t0).count() / reps);
    hipGraph_t g; hipGraphExec_t ge;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { std::printf("begin capture: %s\n", hipGetErrorString(e)); return 1; }
    e = chain();
    hipError_t e2 = hipStreamEndCapture(s, &g);
    if (e != hipSuccess || e2 != hipSuccess) { std::printf("capture failed: chain %s, end %s\n", hipGetErrorString(e), hipGetErrorString(e2)); return 1; }
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; w++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    t2 = std::chrono::steady_clock::now();
    std::printf("graph (%zu nodes): host issue %.1f us per chain, wall %.1f us per chain\n", nn, 1e6 * std::chrono::duration<double>(t1 - t0).count() / reps, 1e6 * std::chrono::duration<double>(t2 - t0).count() / reps);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    t2 = std::chrono::steady_clock::now();
    std::printf("graph + sync each: wall %.1f us per chain\n", 1e6 * std::chrono::duration<double>(t2 - t0).count() / reps);
    return 0;
}
