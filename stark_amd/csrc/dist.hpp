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
//   IpcCollective   : one process per rank, device memory windows shared with hipIpcGetMemHandle / hipIpcOpenMemHandle. A rank's values are
//                     PUSHED into every rank's window by the producing kernel as 8-byte {tag, 32 data bits} granules (system-scope
//                     write-through stores over xGMI; a double is two granules) and the consuming kernel polls its OWN window until every
//                     granule carries the tag of the exchange: no library call, no host involvement, no separate flag or fence — an
//                     exchange costs the one-way store latency. Works with every rank on its own GPU (the peers' windows are mapped
//                     through the xGMI aperture) and with all ranks on ONE device (how the one-GPU test box exercises the real multi-process
//                     launch path of bench.py; RCCL refuses two ranks on one device). Polls are bounded: a peer that never delivers
//                     sets an error word and ends the kernel instead of hanging the GPU.
// All deliver the same bytes to every rank, so reductions done in rank order give every rank identical bits.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <memory>
#include <mutex>
#include <vector>

namespace mistark {

constexpr int MAX_IPC_RANKS = 16;
// What kernels that exchange through the windows themselves need (the fused iteration of the sharded PCG, kernels.hip); see dist.hip
// for the layout. All slots are addressed in granules (8 bytes).
struct IpcView
{
    unsigned long long* win[MAX_IPC_RANKS];  // every rank's window as mapped in this process (win[rank] is the own one)
    int rank = 0, world = 1;
    size_t fast_off = 0;         // first granule of the region reserved for fused kernels
    size_t fast_granules = 0;    // its size
    unsigned int* err = nullptr; // device-visible error word (pinned host memory): != 0 after a poll gave up
    unsigned long long timeout_ticks = 0;  // poll budget on the device's constant clock
    // tags the fused kernels have used up in the fast region (host memory of the communicator): the windows outlive an engine context, so the
    // NEXT context on the same communicator must continue behind the last tag a previous one wrote — a context that started again at tag 1
    // would find its own tags in granules the previous context left there and take stale data for delivered
    uint32_t* fast_tag = nullptr;
};
struct Collective
{
    virtual ~Collective() = default;
    // != nullptr: the ranks exchange through IPC windows and kernels may push / poll themselves
    virtual const IpcView* ipc() { return nullptr; }
    // throws when a bounded wait of an earlier exchange gave up (checked at the host's synchronisation points)
    virtual void check() {}
    // recv[r * n + i] = rank r's send[i], on every rank (send != recv)
    virtual void allgather_f64(const double* send, double* recv, size_t n, hipStream_t stream) = 0;
    // != nullptr: every context of the group must run on this stream (LocalCollective)
    virtual hipStream_t shared_stream() { return nullptr; }
    // 1 = in-process group, 2 = RCCL, 3 = IPC windows; and the number of ranks the transport itself reports (RCCL: ncclCommCount of the
    // communicator; the others: the world they were created with) — what mistark_dist_info hands to the launcher's bench line
    virtual int transport_id() const = 0;
    virtual int transport_ranks() const = 0;
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
// IPC windows: create (allocates and zeroes the window, returns its 64-byte handle), exchange the handles through the launcher, connect.
struct IpcComm;
std::shared_ptr<IpcComm> ipc_comm_create(int device, int rank, int world, size_t window_bytes, char handle_out[64]);
void ipc_comm_connect(IpcComm& comm, const char* handles /* world x 64 bytes, in rank order */);
std::unique_ptr<Collective> make_ipc_collective(std::shared_ptr<IpcComm> comm);
int ipc_comm_rank(const IpcComm& comm);
int ipc_comm_world(const IpcComm& comm);
int ipc_comm_device(const IpcComm& comm);
void rccl_unique_id(char out[128]);
// dist.hip: pre-flight of the windows (one tagged granule over every ordered pair of ranks, bounded by timeout_s <= 2 s) and the RCCL leg
// of the N-GPU bench line (ncclAllReduce of n_big and of 3 doubles)
int ipc_comm_preflight(IpcComm& comm, int iters, double timeout_s, double* half_rtt_us /* world */);
void rccl_allreduce_bench(int device, int rank, int world, const char unique_id[128], size_t n_big, int reps, double out[4]);

}  // namespace mistark
