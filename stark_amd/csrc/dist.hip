// dist.hip — transports of the all-gather every exchange of the sharded path is built from (see dist.hpp)
#include "dist.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <string>

#include "engine.hpp"
#include "ipc_dev.hpp"

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
    int transport_id() const override { return 1; }
    int transport_ranks() const override { return g->world; }
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
    int (*CommCount)(void*, int*) = nullptr;
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
        r.CommCount = (int (*)(void*, int*))sym("ncclCommCount");
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
    int transport_id() const override { return 2; }
    int transport_ranks() const override
    {
        int n = 0;
        nccl_check(rccl().CommCount(comm, &n), "ncclCommCount");
        return n;
    }
    // ncclDataType_t: ncclFloat64 = 8 (rccl.h)
    void allgather_f64(const double* send, double* recv, size_t n, hipStream_t s) override
    {
        if (n) nccl_check(rccl().AllGather(send, recv, n, 8, comm, s), "ncclAllGather(f64)");
    }
};
}  // namespace

// ---- IPC windows (one process per rank; see dist.hpp and ipc_dev.hpp) ---------------------------------------------------------------
// Window of a rank, in granules: [general region | fast region]. General region: slots [parity][source rank][cap doubles x 2 granules] for
// allgather_f64; fast region: laid out by the kernels that exchange there themselves (the sharded PCG's fused iteration).
// Slot reuse. The exchange with sequence number s writes the slots of parity s & 1, so s + 2 overwrites what s delivered. Rank A pushes
// s + 2 after its own wait for s + 1 has finished (stream order), i.e. after EVERY rank B has pushed s + 1, which B enqueued behind its
// wait for s: when a granule of s + 2 lands on B, B has consumed s.
constexpr size_t IPC_HDR = 8;                         // granules at the start of every window: {magic, granules, world, rank}
constexpr unsigned long long IPC_MAGIC = 0x6d69737461726b31ull;  // "mistark1"
struct IpcComm
{
    int device = 0, rank = 0, world = 1;
    size_t granules = 0;       // size of every window
    size_t cap = 0;            // doubles per slot of the general region
    unsigned long long* mine = nullptr;
    bool connected = false;
    std::vector<void*> opened; // peers' windows as returned by hipIpcOpenMemHandle (nullptr for the own rank)
    IpcView view;
    unsigned int* err = nullptr;  // pinned, device-visible
    uint32_t seq = 0;          // exchanges issued so far in the general region
    uint32_t fast_tag = 0;     // tags used up in the fast region (IpcView::fast_tag)
    ~IpcComm()
    {
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();  // (nothing of ours may still poll or push)
        for (void* p : opened)
            if (p) (void)hipIpcCloseMemHandle(p);
        if (mine) (void)hipFree(mine);
        if (err) (void)hipHostFree(err);
    }
};
namespace {
struct IpcPeers
{
    unsigned long long* win[MAX_IPC_RANKS];
};
constexpr int IPC_TB = 256;
// send[i] as two granules into slot `slot` (a granule offset) of every rank's window
__global__ __launch_bounds__(IPC_TB) void k_ipc_push(const double* __restrict__ send, size_t n, IpcPeers pk, int world, size_t slot, uint32_t tag)
{
    for (size_t i = (size_t)blockIdx.x * IPC_TB + threadIdx.x; i < n; i += (size_t)gridDim.x * IPC_TB) {
        const double v = send[i];
        for (int r = 0; r < world; r++) granule_store_f64(pk.win[r] + slot + 2 * i, tag, v);
    }
}
// recv[r * n + i] = what rank r pushed, taken from the own window once both granules carry the tag
__global__ __launch_bounds__(IPC_TB) void k_ipc_wait(const unsigned long long* __restrict__ mine, size_t slot0, size_t slot_granules, int world, size_t n, uint32_t tag,
                                                     double* __restrict__ recv, unsigned int* err, unsigned long long budget)
{
    const unsigned long long t0 = wall_clock64();
    const size_t total = (size_t)world * n;
    for (size_t t = (size_t)blockIdx.x * IPC_TB + threadIdx.x; t < total; t += (size_t)gridDim.x * IPC_TB) {
        const size_t r = t / n, i = t - r * n;
        recv[t] = granule_wait_f64(mine + slot0 + r * slot_granules + 2 * i, tag, err, t0, budget, 1u | (tag << 8));
    }
}
struct IpcCollective : Collective
{
    std::shared_ptr<IpcComm> m;
    explicit IpcCollective(std::shared_ptr<IpcComm> comm) : m(std::move(comm))
    {
        if (!m->connected) throw Error("IPC communicator: connect it before use (mistark_ipc_comm_connect)");
    }
    const IpcView* ipc() override { return &m->view; }
    int transport_id() const override { return 3; }
    int transport_ranks() const override { return m->world; }
    void check() override
    {
        const unsigned int code = __atomic_load_n(m->err, __ATOMIC_ACQUIRE);
        if (code != 0)
            throw Error("multi-GPU (IPC windows): an exchange gave up waiting for a peer (rank " + std::to_string(m->rank) + " of " + std::to_string(m->world) + ", wait code " +
                        std::to_string(code & 0xffu) + ", iteration " + std::to_string(code >> 8) +
                        "): a peer process failed, or the ranks did not issue the same sequence of exchanges");
    }
    void allgather_f64(const double* send, double* recv, size_t n, hipStream_t s) override
    {
        if (n == 0) return;
        check();  // (an earlier exchange that gave up: fail here, not many exchanges later)
        static const bool dbg = std::getenv("MISTARK_DEBUG_FUSED") != nullptr;
        if (dbg) std::fprintf(stderr, "[ipc r%d] all-gather %u of %zu doubles\n", m->rank, m->seq + 1, n);
        const int W = m->world;
        IpcPeers pk{};
        for (int r = 0; r < W; r++) pk.win[r] = m->view.win[r];
        const size_t slot_g = 2 * m->cap;
        // (messages longer than a slot travel in pieces, each its own exchange; recv is laid out per rank with stride n)
        if (n <= m->cap) {
            if (m->seq == 0xffffffffu) throw Error("IPC communicator: 2^32 exchanges issued; the tag would wrap to the windows' zero-filled state (create a new communicator)");
            const uint32_t tag = ++m->seq;
            const size_t par = IPC_HDR + (size_t)(tag & 1u) * (size_t)W * slot_g;
            const int gp = (int)std::min<size_t>((n + IPC_TB - 1) / IPC_TB, 256);
            hipLaunchKernelGGL(k_ipc_push, dim3(gp), dim3(IPC_TB), 0, s, send, n, pk, W, par + (size_t)m->rank * slot_g, tag);
            // the waiting grid stays small: polling workgroups must never crowd out the kernels they are waiting for when several ranks
            // share one device
            const int gw = (int)std::min<size_t>(((size_t)W * n + IPC_TB - 1) / IPC_TB, 64);
            hipLaunchKernelGGL(k_ipc_wait, dim3(gw), dim3(IPC_TB), 0, s, (const unsigned long long*)m->mine, par, slot_g, W, n, tag, recv, m->err, m->view.timeout_ticks);
            return;
        }
        for (size_t at = 0; at < n; at += m->cap) {
            const size_t len = std::min(m->cap, n - at);
            ensure_tmp((size_t)W * len);
            allgather_f64(send + at, chunk_tmp, len, s);
            for (int r = 0; r < W; r++) MS_CHECK(hipMemcpyAsync(recv + (size_t)r * n + at, chunk_tmp + (size_t)r * len, len * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
    }
    double* chunk_tmp = nullptr;  // scratch of the pieces (allocated on demand)
    size_t chunk_tmp_n = 0;
    void ensure_tmp(size_t n)
    {
        if (chunk_tmp && chunk_tmp_n >= n) return;
        if (chunk_tmp) (void)hipFree(chunk_tmp);
        MS_CHECK(hipMalloc((void**)&chunk_tmp, n * sizeof(double)));
        chunk_tmp_n = n;
    }
    ~IpcCollective() override
    {
        if (chunk_tmp) (void)hipFree(chunk_tmp);
    }
};
}  // namespace

// header (IPC_HDR granules: what the peers check at connect) | general region: 3/4 of the window, 2 parities x world slots | the rest for
// kernels that exchange by themselves
static void ipc_layout(IpcComm& m)
{
    const size_t fast = m.granules / 4;
    const size_t gen = m.granules - fast - IPC_HDR;
    m.cap = gen / (2 * (size_t)m.world * 2);
    m.view.rank = m.rank;
    m.view.world = m.world;
    m.view.fast_off = m.granules - fast;
    m.view.fast_granules = fast;
    m.view.err = m.err;
    m.view.fast_tag = &m.fast_tag;
    int khz = 0;
    MS_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m.device));
    double seconds = 30.0;
    if (const char* env = std::getenv("MISTARK_IPC_TIMEOUT_S")) seconds = std::max(0.05, std::atof(env));
    m.view.timeout_ticks = (unsigned long long)(seconds * 1e3 * (double)std::max(khz, 1000));
}
std::shared_ptr<IpcComm> ipc_comm_create(int device, int rank, int world, size_t window_bytes, char handle_out[64])
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
    if (world < 1 || world > MAX_IPC_RANKS || rank < 0 || rank >= world) throw Error("IPC communicator: rank / world out of range (at most " + std::to_string(MAX_IPC_RANKS) + " ranks)");
    auto m = std::make_shared<IpcComm>();
    m->device = device;
    m->rank = rank;
    m->world = world;
    MS_CHECK(hipSetDevice(device));
    const size_t min_bytes = (size_t)world * 64 * 1024;
    m->granules = std::max(window_bytes, min_bytes) / 8;
    // Uncached device memory (what RCCL allocates its own peer buffers with): stores from other GPUs land in HBM behind this GPU's L2, so the
    // local mapping must not keep lines in it. When the runtime refuses to export such an allocation, plain device memory still serves ranks
    // that share ONE device (system-scope accesses on both sides).
    hipError_t e = hipExtMallocWithFlags((void**)&m->mine, m->granules * 8, hipDeviceMallocUncached);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, m->mine);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (m->mine) (void)hipFree(m->mine);
        m->mine = nullptr;
        MS_CHECK(hipMalloc((void**)&m->mine, m->granules * 8));
        MS_CHECK(hipIpcGetMemHandle(&h, m->mine));
    }
    MS_CHECK(hipMemset(m->mine, 0, m->granules * 8));
    // every rank derives the slot offsets from ITS window size and world: the header lets the peers refuse a mismatch at connect instead of
    // misdelivering or timing out later
    const unsigned long long hdr[4] = {IPC_MAGIC, (unsigned long long)m->granules, (unsigned long long)world, (unsigned long long)rank};
    MS_CHECK(hipMemcpy(m->mine, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    MS_CHECK(hipDeviceSynchronize());  // zeroed and labelled before anybody can learn the handle
    MS_CHECK(hipHostMalloc((void**)&m->err, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *m->err = 0;
    std::memcpy(handle_out, &h, 64);
    ipc_layout(*m);
    return m;
}
void ipc_comm_connect(IpcComm& m, const char* handles)
{
    if (m.connected) throw Error("IPC communicator: already connected");
    MS_CHECK(hipSetDevice(m.device));
    m.opened.assign((size_t)m.world, nullptr);
    for (int r = 0; r < m.world; r++) {
        if (r == m.rank) {
            m.view.win[r] = m.mine;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        MS_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        m.opened[(size_t)r] = p;
        m.view.win[r] = (unsigned long long*)p;
        unsigned long long hdr[4] = {0, 0, 0, 0};
        MS_CHECK(hipMemcpy(hdr, p, sizeof(hdr), hipMemcpyDeviceToHost));
        if (hdr[0] != IPC_MAGIC || hdr[1] != (unsigned long long)m.granules || hdr[2] != (unsigned long long)m.world || hdr[3] != (unsigned long long)r)
            throw Error("IPC communicator: rank " + std::to_string(r) + "'s window does not match this rank's layout (window of " + std::to_string(hdr[1] * 8) + " bytes, world " +
                        std::to_string(hdr[2]) + ", rank " + std::to_string(hdr[3]) + "; here " + std::to_string(m.granules * 8) + " bytes, world " + std::to_string(m.world) +
                        "): every rank must create its communicator with the same window size and world, handles in rank order");
    }
    m.connected = true;
}
std::unique_ptr<Collective> make_ipc_collective(std::shared_ptr<IpcComm> comm) { return std::make_unique<IpcCollective>(std::move(comm)); }
int ipc_comm_rank(const IpcComm& comm) { return comm.rank; }
int ipc_comm_world(const IpcComm& comm) { return comm.world; }
int ipc_comm_device(const IpcComm& comm) { return comm.device; }

// (Ranks inside one process cannot use the windows: HIP maps the streams of a process onto a few hardware queues, and a polling kernel of one
// rank that shares a queue with the stream of the rank it waits for blocks that rank's push — measured: the in-process variant dead-locked
// until the poll's time-out. Separate processes own separate queues.)
std::unique_ptr<Collective> make_local_collective(std::shared_ptr<LocalGroup> group, int rank, int device) { return std::make_unique<LocalCollective>(std::move(group), rank, device); }
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, const char uid[128]) { return std::make_unique<RcclCollective>(rank, world, uid); }
void rccl_unique_id(char out[128])
{
    Id id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, 128);
}

}  // namespace mistark
