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
constexpr size_t IPC_PRE = 2 * MAX_IPC_RANKS;         // granules of the pre-flight's ping / pong slots (one each per source rank), behind the general region
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
    size_t pre_off = 0;        // first granule of the pre-flight's slots
    uint32_t pre_tag = 0;      // tags the pre-flights have used up
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

// ---- pre-flight: one tagged granule over every ordered pair of ranks, with a short time-out -----------------------------------------------
// What a first run on N real GPUs has to find out before anything is built on the windows: does a system-scope store into a peer's mapped
// window ARRIVE (over xGMI, or inside one device when the ranks share it), and how long does it take. One lane per rank plays ping-pong with
// every peer in turn: in round k rank r pings p = r + k (mod W) and answers the ping of q = r - k; `iters` exchanges per round, each with
// its own tag; the round trip is taken on the device's constant clock around ping store -> pong seen. Every wait is bounded by `budget`
// ticks counted from the kernel's start: a granule that never arrives ends the kernel with rtt = 0 for that peer instead of hanging.
// Slots (own region of the window, dist.hip "ipc_layout"): [pre_off + s] = ping of source s, [pre_off + MAX_IPC_RANKS + s] = pong of source s.
__device__ __forceinline__ bool granule_poll_tag(const unsigned long long* g, uint32_t tag, unsigned long long t0, unsigned long long budget)
{
    for (unsigned spins = 0;; spins++) {
        if ((uint32_t)(granule_load(g) >> 32) == tag) return true;
        if ((spins & 63u) == 63u && wall_clock64() - t0 > budget) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
__global__ void k_ipc_preflight(IpcPeers pk, int rank, int world, size_t pre_off, uint32_t tag0, int iters, unsigned long long budget, unsigned long long* rtt_min /* world */,
                                unsigned int* failed /* 0, or 1 + the peer whose granule did not arrive */)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t_start = wall_clock64();
    const unsigned long long* mine = pk.win[rank];
    for (int k = 1; k < world; k++) {
        const int p = (rank + k) % world, q = (rank - k + world) % world;
        unsigned long long best = ~0ull;
        for (int it = 0; it < iters; it++) {
            const uint32_t tag = tag0 + (uint32_t)((k - 1) * iters + it) + 1u;
            const unsigned long long t0 = wall_clock64();
            granule_store(pk.win[p] + pre_off + (size_t)rank, tag, (uint32_t)rank);                         // ping p
            if (!granule_poll_tag(mine + pre_off + (size_t)q, tag, t_start, budget)) { *failed = 1u + (unsigned)q; return; }  // q's ping
            granule_store(pk.win[q] + pre_off + MAX_IPC_RANKS + (size_t)rank, tag, (uint32_t)rank);         // pong to q
            if (!granule_poll_tag(mine + pre_off + MAX_IPC_RANKS + (size_t)p, tag, t_start, budget)) { *failed = 1u + (unsigned)p; return; }  // p's pong
            const unsigned long long dt = wall_clock64() - t0;
            if (it > 0 && dt < best) best = dt;  // (the first exchange of a round holds the ranks' skew)
        }
        rtt_min[p] = best;
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
    const size_t gen = m.granules - fast - IPC_HDR - IPC_PRE;
    m.cap = gen / (2 * (size_t)m.world * 2);
    m.pre_off = m.granules - fast - IPC_PRE;
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

// Collective (every rank, same arguments; the caller puts a host barrier in front so that the kernels start within the time-out of each other).
// half_rtt_us[p] = half the best round trip with peer p in microseconds (0 for the own rank, < 0 where no granule came back);
// returns the number of peers that answered.
int ipc_comm_preflight(IpcComm& m, int iters, double timeout_s, double* half_rtt_us)
{
    if (!m.connected) throw Error("IPC communicator: connect it before the pre-flight");
    if (iters < 2) iters = 2;
    MS_CHECK(hipSetDevice(m.device));
    int khz = 0;
    MS_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m.device));
    khz = std::max(khz, 1000);
    const unsigned long long budget = (unsigned long long)(std::min(std::max(timeout_s, 0.01), 2.0) * 1e3 * (double)khz);
    const int W = m.world;
    if ((unsigned long long)m.pre_tag + (unsigned long long)W * (unsigned long long)iters + 1ull > 0xffffffffull) throw Error("IPC communicator: pre-flight tags used up");
    IpcPeers pk{};
    for (int r = 0; r < W; r++) pk.win[r] = m.view.win[r];
    // (test hook: MISTARK_IPC_FAULT=drop_preflight makes rank 1's stores land in its OWN window instead of the peers' — what a mapping that does
    // not reach the peer looks like from the outside: nothing ever arrives, and the pre-flight has to notice within its time-out)
    if (const char* f = std::getenv("MISTARK_IPC_FAULT"))
        if (std::strcmp(f, "drop_preflight") == 0 && m.rank == 1)
            for (int r = 0; r < W; r++) pk.win[r] = m.mine - (r == m.rank ? 0 : IPC_PRE);  // (the general region's unused tail: the communicator is dropped afterwards)
    unsigned long long* res = nullptr;  // [world] round trips in ticks | failed word
    MS_CHECK(hipHostMalloc((void**)&res, (MAX_IPC_RANKS + 1) * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i <= MAX_IPC_RANKS; i++) res[i] = 0;
    hipStream_t s;
    MS_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_ipc_preflight, dim3(1), dim3(64), 0, s, pk, m.rank, W, m.pre_off, m.pre_tag, iters, budget, res, (unsigned int*)(res + MAX_IPC_RANKS));
    const hipError_t e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    m.pre_tag += (uint32_t)((W - 1) * iters);
    int ok = 0;
    for (int p = 0; p < W; p++) {
        if (p == m.rank) half_rtt_us[p] = 0.0;
        else if (res[p] == 0 || res[p] == ~0ull) half_rtt_us[p] = -1.0;
        else {
            half_rtt_us[p] = 0.5 * (double)res[p] / (double)khz * 1e3;
            ok++;
        }
    }
    (void)hipHostFree(res);
    MS_CHECK(e);
    return ok;
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

// ---- RCCL leg of the N-GPU bench line: the two all-reduces north_star names, timed on THIS transport whatever the default one is ----------
// ncclAllReduce (sum, f64) of `n_big` doubles — the gradient of the shared DoFs (SymX's reduction of the thread-local gradients,
// SecondOrderCompiledGlobal.cpp:72-142) — and of 3 doubles — the dot products of one CG iteration (BlockedSparseMatrix/solve_pcg.h:180,201,217)
// — `reps` launches each, back to back on one stream between HIP events, after 5 warm-up launches; every element of the first result is
// checked against the closed-form sum over the ranks. out[0] = ncclCommCount of the communicator, out[1] = microseconds per all-reduce of
// n_big doubles, out[2] = of 3 doubles, out[3] = seconds ncclCommInitRank took. Collective: every rank calls it with the same arguments and
// rank 0's unique id; one rank per device (RCCL refuses two ranks on one).
void rccl_allreduce_bench(int device, int rank, int world, const char uid[128], size_t n_big, int reps, double out[4])
{
    if (world < 1 || rank < 0 || rank >= world || n_big == 0 || reps <= 0) throw Error("rccl_allreduce_bench: bad arguments");
    MS_CHECK(hipSetDevice(device));
    const auto t_init = std::chrono::steady_clock::now();
    RcclCollective coll(rank, world, uid);
    out[3] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_init).count();
    out[0] = (double)coll.transport_ranks();
    hipStream_t s;
    MS_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    DevBuf<double> send, recv;
    send.ensure(n_big);
    recv.ensure(n_big);
    std::vector<double> h(n_big);
    for (size_t i = 0; i < n_big; i++) h[i] = (double)(rank + 1) * (double)(1 + i % 7);  // (small integers: the sum is exact in any order)
    MS_CHECK(hipMemcpy(send.p, h.data(), n_big * sizeof(double), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    MS_CHECK(hipEventCreate(&e0));
    MS_CHECK(hipEventCreate(&e1));
    const double ranks_sum = 0.5 * (double)world * (double)(world + 1);
    int slot = 1;
    for (size_t n : {n_big, (size_t)3}) {
        MS_CHECK(hipMemsetAsync(recv.p, 0, n * sizeof(double), s));
        for (int w = 0; w < 5; w++) nccl_check(rccl().AllReduce(send.p, recv.p, n, 8 /* ncclFloat64 */, 0 /* ncclSum */, coll.comm, s), "ncclAllReduce(f64, sum)");
        MS_CHECK(hipMemcpyAsync(h.data(), recv.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        MS_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; i++)
            if (h[i] != ranks_sum * (double)(1 + i % 7))
                throw Error("RCCL all-reduce of " + std::to_string(n) + " doubles over " + std::to_string(world) + " ranks: entry " + std::to_string(i) + " is " + std::to_string(h[i]) +
                            ", expected " + std::to_string(ranks_sum * (double)(1 + i % 7)));
        MS_CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) nccl_check(rccl().AllReduce(send.p, recv.p, n, 8, 0, coll.comm, s), "ncclAllReduce(f64, sum)");
        MS_CHECK(hipEventRecord(e1, s));
        MS_CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        MS_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out[slot++] = 1e3 * (double)ms / (double)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    MS_CHECK(hipStreamSynchronize(s));
    (void)hipStreamDestroy(s);
}

}  // namespace mistark
