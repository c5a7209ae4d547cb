// dist.hpp — sum all-reduce over the ranks that share one sharded problem.
//
// Two transports behind one interface:
//   RcclCollective  : one process per GPU; RCCL (librccl, loaded with dlopen at first use) over xGMI, enqueued on the engine's stream.
//   LocalCollective : several engine contexts inside ONE process (one host thread each), any mix of devices, synchronised with a
//                     barrier; used by the tests to run the N > 1 path on a single MI355X, and usable for single-process multi-GPU.
// Both give every rank the bit-identical sum (the replicated parts of the solver rely on that).
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
    virtual void allreduce_f64(double* buf, size_t n, hipStream_t stream) = 0;
    virtual void allreduce_f32(float* buf, size_t n, hipStream_t stream) = 0;
};

// contiguous element ranges: elements [n*rank/world, n*(rank+1)/world)
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
    std::vector<void*> ptr;
    explicit LocalGroup(int w) : world(w), ptr((size_t)w, nullptr) {}
    void barrier();
};
std::unique_ptr<Collective> make_local_collective(std::shared_ptr<LocalGroup> group, int rank);
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, const char unique_id[128]);
void rccl_unique_id(char out[128]);

}  // namespace mistark
