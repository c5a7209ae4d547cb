// dist.hip — transports of the sum all-reduce (see dist.hpp)
#include "dist.hpp"

#include <dlfcn.h>

#include <cstring>
#include <string>

#include "engine.hpp"

namespace mistark {

void LocalGroup::barrier()
{
    std::unique_lock<std::mutex> lk(m);
    const long long gen = generation;
    if (++arrived == world) {
        arrived = 0;
        generation++;
        cv.notify_all();
    } else {
        cv.wait(lk, [&] { return generation != gen; });
    }
}

struct Id  // ncclUniqueId (rccl.h:43)
{
    char internal[128];
};
namespace {
constexpr int MAX_LOCAL = 16;
template <class T>
struct PtrPack
{
    const T* p[MAX_LOCAL];
};
template <class T>
__global__ void k_sum_ranks(PtrPack<T> in, int world, size_t n, T* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T s = in.p[0][i];
        for (int r = 1; r < world; r++) s += in.p[r][i];  // fixed order: every rank computes the same bits
        out[i] = s;
    }
}
struct LocalCollective : Collective
{
    std::shared_ptr<LocalGroup> g;
    int rank;
    DevBuf<double> tmp;
    LocalCollective(std::shared_ptr<LocalGroup> group, int r) : g(std::move(group)), rank(r)
    {
        if (g->world > MAX_LOCAL) throw Error("local group too large");
    }
    template <class T>
    void run(T* buf, size_t n, hipStream_t stream)
    {
        if (n == 0) return;
        MS_CHECK(hipStreamSynchronize(stream));  // my contribution is complete
        g->ptr[(size_t)rank] = buf;
        g->barrier();
        PtrPack<T> pk{};
        for (int r = 0; r < g->world; r++) pk.p[r] = (const T*)g->ptr[(size_t)r];
        tmp.ensure((n * sizeof(T) + sizeof(double) - 1) / sizeof(double));
        const int grid = (int)std::min<size_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(k_sum_ranks<T>, dim3(grid), dim3(256), 0, stream, pk, g->world, n, (T*)tmp.p);
        MS_CHECK(hipStreamSynchronize(stream));
        g->barrier();  // everybody has read everybody's contribution
        MS_CHECK(hipMemcpyAsync(buf, tmp.p, n * sizeof(T), hipMemcpyDeviceToDevice, stream));
        MS_CHECK(hipStreamSynchronize(stream));
        g->barrier();
    }
    void allreduce_f64(double* buf, size_t n, hipStream_t s) override { run(buf, n, s); }
    void allreduce_f32(float* buf, size_t n, hipStream_t s) override { run(buf, n, s); }
};

// ---- RCCL through dlopen: no link-time dependency, and no clash with a librccl another module of the process brought ---------
struct Rccl
{
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id, int) = nullptr;  // ncclUniqueId is passed by value
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl()
{
    static Rccl r;
    if (!r.lib) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) throw Error("multi-GPU: cannot load librccl.so");
        auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p) throw Error(std::string("multi-GPU: librccl lacks ") + n);
            return p;
        };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, Id, int))sym("ncclCommInitRank");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    }
    return r;
}
void nccl_check(int rc, const char* what)
{
    if (rc != 0) throw Error(std::string("RCCL ") + what + ": " + rccl().GetErrorString(rc));
}
struct RcclCollective : Collective
{
    void* comm = nullptr;
    RcclCollective(int rank, int world, const char uid[128])
    {
        Id id;
        std::memcpy(id.internal, uid, 128);
        nccl_check(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    }
    ~RcclCollective() override
    {
        if (comm) rccl().CommDestroy(comm);
    }
    // ncclDataType_t: ncclFloat32 = 7, ncclFloat64 = 8; ncclRedOp_t: ncclSum = 0 (rccl.h)
    void allreduce_f64(double* buf, size_t n, hipStream_t s) override
    {
        if (n) nccl_check(rccl().AllReduce(buf, buf, n, 8, 0, comm, s), "ncclAllReduce(f64)");
    }
    void allreduce_f32(float* buf, size_t n, hipStream_t s) override
    {
        if (n) nccl_check(rccl().AllReduce(buf, buf, n, 7, 0, comm, s), "ncclAllReduce(f32)");
    }
};
}  // namespace

std::unique_ptr<Collective> make_local_collective(std::shared_ptr<LocalGroup> group, int rank) { return std::make_unique<LocalCollective>(std::move(group), rank); }
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, const char uid[128]) { return std::make_unique<RcclCollective>(rank, world, uid); }
void rccl_unique_id(char out[128])
{
    Id id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, 128);
}

}  // namespace mistark
