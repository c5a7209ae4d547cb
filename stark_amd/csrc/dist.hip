// dist.hip — transports of the all-gather every exchange of the sharded path is built from (see dist.hpp)
#include "dist.hpp"

#include <dlfcn.h>

#include <chrono>
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
        // (a rank that failed never arrives: give up instead of hanging the process)
        if (!cv.wait_for(lk, std::chrono::seconds(180), [&] { return generation != gen; })) throw Error("in-process group: a rank did not reach the exchange (did it fail?)");
    }
}
LocalGroup::~LocalGroup()
{
    if (stream) (void)hipStreamDestroy(stream);
}

struct Id  // ncclUniqueId (rccl.h:43)
{
    char internal[128];
};
namespace {
constexpr int MAX_LOCAL = 16;
struct GatherPack
{
    const double* send[MAX_LOCAL];
    double* recv[MAX_LOCAL];
};
// every rank's recv[r * n + i] = rank r's send[i]
__global__ void k_allgather_local(GatherPack pk, int world, size_t n)
{
    const size_t total = (size_t)world * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(t / n);
        const double v = pk.send[r][t - (size_t)r * n];
        for (int q = 0; q < world; q++) pk.recv[q][t] = v;
    }
}
struct LocalCollective : Collective
{
    std::shared_ptr<LocalGroup> g;
    int rank;
    LocalCollective(std::shared_ptr<LocalGroup> group, int r, int device) : g(std::move(group)), rank(r)
    {
        if (g->world > MAX_LOCAL) throw Error("local group too large");
        std::lock_guard<std::mutex> lk(g->m);
        if (!g->stream) {
            MS_CHECK(hipSetDevice(device));
            MS_CHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
            g->device = device;
        } else if (g->device != device) {
            throw Error("in-process group: every rank must use the same device (the ranks share one stream)");
        }
    }
    hipStream_t shared_stream() override { return g->stream; }
    void allgather_f64(const double* send, double* recv, size_t n, hipStream_t stream) override
    {
        if (n == 0) return;
        g->send[(size_t)rank] = send;
        g->recv[(size_t)rank] = recv;
        g->barrier();  // every rank has enqueued what produces its `send` (same stream: ordered before the kernel below)
        if (rank == 0) {
            GatherPack pk{};
            for (int r = 0; r < g->world; r++) {
                pk.send[r] = (const double*)g->send[(size_t)r];
                pk.recv[r] = (double*)g->recv[(size_t)r];
            }
            const int grid = (int)std::min<size_t>(((size_t)g->world * n + 255) / 256, 1024);
            hipLaunchKernelGGL(k_allgather_local, dim3(grid), dim3(256), 0, stream, pk, g->world, n);
        }
        g->barrier();  // the copy is enqueued: what the ranks enqueue from here on runs after it
    }
};

// ---- RCCL through dlopen: no link-time dependency, and no clash with a librccl another module of the process brought ---------
struct Rccl
{
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id, int) = nullptr;  // ncclUniqueId is passed by value
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
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
        r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
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
    // ncclDataType_t: ncclFloat64 = 8 (rccl.h)
    void allgather_f64(const double* send, double* recv, size_t n, hipStream_t s) override
    {
        if (n) nccl_check(rccl().AllGather(send, recv, n, 8, comm, s), "ncclAllGather(f64)");
    }
};
}  // namespace

std::unique_ptr<Collective> make_local_collective(std::shared_ptr<LocalGroup> group, int rank, int device) { return std::make_unique<LocalCollective>(std::move(group), rank, device); }
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, const char uid[128]) { return std::make_unique<RcclCollective>(rank, world, uid); }
void rccl_unique_id(char out[128])
{
    Id id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, 128);
}

}  // namespace mistark
