// dist.hpp — the exchange primitive of a sharded problem and its transports.
//
// Everything the sharded path exchanges is built from ONE collective, an all-gather of n doubles per rank (shard.hip: scalars such
// as energies and dot products, boundary values of vectors, the owned parts of a solution). Transports:
//   RcclCollective  : one process per GPU; ncclAllGather (librccl, loaded with dlopen at first use) over xGMI, enqueued on the engine's
//                     stream (stream-ordered: no host synchronisation).
//   LocalCollective : several engine contexts inside ONE process, one host thread each, all on one device and ONE shared HIP stream
//                     (the group's): the ranks' host threads meet at a barrier once their producers are enqueued, one of them enqueues
//                     the copy kernel, and stream order does the rest: nothing ever waits on the GPU. Used by the tests to run the N > 1
//                     path (2, 3, 8 ranks) on a single MI355X.
// Both deliver the same bytes to every rank, so reductions done in rank order give every rank identical bits.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <memory>
#include <mutex>
#include <vector>

namespace mistark {

struct Collective
{
    virtual ~Collective() = default;
    // recv[r * n + i] = rank r's send[i], on every rank (send != recv)
    virtual void allgather_f64(const double* send, double* recv, size_t n, hipStream_t stream) = 0;
    // != nullptr: every context of the group must run on this stream (LocalCollective)
    virtual hipStream_t shared_stream() { return nullptr; }
};

// contiguous ranges: [n*rank/world, n*(rank+1)/world)
inline void shard_range(long long n, int rank, int world, long long& begin, long long& end)
{
    begin = n * rank / world;
    end = n * (rank + 1) / world;
}

struct LocalGroup
{
    int world;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    std::vector<const void*> send;
    std::vector<void*> recv;
    hipStream_t stream = nullptr;
    int device = -1;
    explicit LocalGroup(int w) : world(w), send((size_t)w, nullptr), recv((size_t)w, nullptr) {}
    ~LocalGroup();
    void barrier();
};
std::unique_ptr<Collective> make_local_collective(std::shared_ptr<LocalGroup> group, int rank, int device);
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, const char unique_id[128]);
void rccl_unique_id(char out[128]);

}  // namespace mistark
