// Micro-benchmark (round 6): symmetric-half block storage for the SpMV — diagonal + upper 3x3 float blocks stored once, a
// row's lower blocks read a second time from where their transposes lie (per-row gather list, fixed order, no atomics) — against the full
// storage, on a synthetic matrix with the structure of configs[3]'s: nodes of an nx x ny x nz grid in Morton order, 14 neighbours per node
// (the tet-grid stencil: +-x, +-y, +-z, +-(1,1,0), +-(0,1,1), +-(1,0,1), +-(1,1,1)) = 15 blocks per interior row, float values, double vectors.
// Both kernels share the row-aligned chunks / lane-per-block / prefix-scan structure of the engine's k_spmv_fused (stark_amd/csrc/solve.hip), so
// the difference between them is the storage scheme, and `full` is calibrated against the engine's own launch (21 us / 187 us).
// Also timed: the plain float4 stream of each value buffer (the floor of the memory system for that many bytes).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 sym_spmv.hip -o sym_spmv.bin ; run: ./sym_spmv.bin 56 56 55 [chunk_tiles] [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)
constexpr int BLOCK = 256;
constexpr uint32_t PAD = 0xFFFFFFFFu;

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, BOUND);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, BOUND);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int addr)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// row sums of the 64 lane values (y0, y1, y2) of a tile whose row ends are marked by `tails`; on return the tail lanes hold their row's sum
// (first segment + carry k from the previous tile; an open last segment is handed on in k). The engine's scan, verbatim in structure.
__device__ __forceinline__ void tile_row_sums(double& y0, double& y1, double& y2, unsigned long long tails, bool tile_cont, int lane, double& k0, double& k1, double& k2)
{
    const unsigned long long heads = (tails << 1) | 1ull;
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int start = 63 - __clzll(heads & le);
    const double v0 = y0, v1 = y1, v2 = y2;
#define SCAN_STEP(CTRL, RM, BOUND)                                                                                            \
    {                                                                                                                         \
        const double u0 = dpp_mov<CTRL, RM, BOUND>(y0), u1 = dpp_mov<CTRL, RM, BOUND>(y1), u2 = dpp_mov<CTRL, RM, BOUND>(y2); \
        y0 += u0; y1 += u1; y2 += u2;                                                                                         \
    }
    SCAN_STEP(0x111, 0xf, true)
    SCAN_STEP(0x112, 0xf, true)
    SCAN_STEP(0x114, 0xf, true)
    SCAN_STEP(0x118, 0xf, true)
    SCAN_STEP(0x142, 0xa, false)
    SCAN_STEP(0x143, 0xc, false)
#undef SCAN_STEP
    const int addr = start << 2;
    const double e0 = y0 - v0, e1 = y1 - v1, e2 = y2 - v2;
    y0 -= lane_gather(e0, addr);
    y1 -= lane_gather(e1, addr);
    y2 -= lane_gather(e2, addr);
    if (tile_cont && start == 0) { y0 += k0; y1 += k1; y2 += k2; }
    if (((tails >> 63) & 1ull) == 0ull) { k0 = read_lane(y0, 63); k1 = read_lane(y1, 63); k2 = read_lane(y2, 63); }
}

struct Part  // one stream of tiles: 64 entries each; word bit 31 = last entry of its row in this stream
{
    const float* vals;         // [ntiles][576]   (stored blocks only; null for the gather stream)
    const uint32_t* word;      // [ntiles][64]    column (stored blocks) / source row i (gathered blocks); PAD = no entry
    const uint32_t* pos;       // [ntiles][64]    gathered blocks: position (tile * 64 + lane) of the transposed block in the stored stream
    const uint32_t* dst;       // [ntiles][64]    destination row of the entry's row sum (read by tail lanes)
    const int32_t* chunk_t0;   // [nchunks + 1]   first tile of a chunk (chunks hold whole rows)
};
// MODE 0: y = A x on the stored blocks (full storage, or the upper half of the symmetric one)
// MODE 1: y += (lower blocks) x: entry (pos, i, j): block B = stored (i, j), contribution B^T x_i to row j
template <int MODE>
__device__ __forceinline__ void run_stream(const Part& P, const float* __restrict__ vals_stored, int64_t ch, int lane, const double* __restrict__ x, double* __restrict__ y)
{
    const int t0 = P.chunk_t0[ch], t1 = P.chunk_t0[ch + 1];
    double k0 = 0.0, k1 = 0.0, k2 = 0.0;
    bool cont = false;
    for (int t = t0; t < t1; t++) {
        const uint32_t w = P.word[(size_t)t * 64 + lane];
        const bool valid = w != PAD;
        const bool tail = valid && (w >> 31) != 0;
        const size_t c3 = 3 * (size_t)(valid ? (w & 0x7fffffffu) : 0u);
        const double x0 = x[c3], x1 = x[c3 + 1], x2 = x[c3 + 2];
        float4 a, b;
        float cc;
        if (MODE == 0) {
            const float4* q = reinterpret_cast<const float4*>(P.vals + (size_t)t * 576);
            a = q[lane];
            b = q[64 + lane];
            cc = P.vals[(size_t)t * 576 + 512 + lane];
        } else {
            const uint32_t p = valid ? P.pos[(size_t)t * 64 + lane] : 0u;
            const float* tv = vals_stored + (size_t)(p >> 6) * 576;
            const int l = (int)(p & 63);
            a = reinterpret_cast<const float4*>(tv)[l];
            b = reinterpret_cast<const float4*>(tv)[64 + l];
            cc = tv[512 + l];
        }
        double y0, y1, y2;
        if (MODE == 0) {
            y0 = (double)a.x * x0 + (double)a.y * x1 + (double)a.z * x2;
            y1 = (double)a.w * x0 + (double)b.x * x1 + (double)b.y * x2;
            y2 = (double)b.z * x0 + (double)b.w * x1 + (double)cc * x2;
        } else {  // transposed product
            y0 = (double)a.x * x0 + (double)a.w * x1 + (double)b.z * x2;
            y1 = (double)a.y * x0 + (double)b.x * x1 + (double)b.w * x2;
            y2 = (double)a.z * x0 + (double)b.y * x1 + (double)cc * x2;
        }
        if (!valid) y0 = y1 = y2 = 0.0;
        const unsigned long long tails = __ballot(tail);
        tile_row_sums(y0, y1, y2, tails, cont, lane, k0, k1, k2);
        cont = ((tails >> 63) & 1ull) == 0ull;
        if (tail) {
            double* yr = y + 3 * (size_t)P.dst[(size_t)t * 64 + lane];
            if (MODE == 0) {
                yr[0] = y0; yr[1] = y1; yr[2] = y2;
            } else {  // the same wavefront stored the row's upper sum in its first pass: plain read-modify-write, fixed order
                yr[0] += y0; yr[1] += y1; yr[2] += y2;
            }
        }
    }
}
template <bool SYM>
__global__ __launch_bounds__(BLOCK) void k_spmv(Part U, Part Lw, int64_t n_chunks, const double* __restrict__ x, double* __restrict__ y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk = (int)gridDim.x, bid = (int)blockIdx.x;
    const int pbid = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;  // an XCD gets a contiguous eighth of the chunks
    const int64_t n_waves = (int64_t)nblk * 4;
    for (int64_t ch = (int64_t)pbid * 4 + wave; ch < n_chunks; ch += n_waves) {
        run_stream<0>(U, U.vals, ch, lane, x, y);
        if (SYM) {
            __builtin_amdgcn_s_waitcnt(0);  // (the pass above stored the rows this pass adds to)
            run_stream<1>(Lw, U.vals, ch, lane, x, y);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_stream(const float4* __restrict__ v, size_t n4, double* __restrict__ out)
{
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLOCK) {
        const float4 a = v[i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.0;
}

// ---- host: matrix and storage ------------------------------------------------------------------------------------------------------
static uint64_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    auto spread = [](uint64_t v) {
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    return spread(x) | spread(y) << 1 | spread(z) << 2;
}
struct HostPart
{
    std::vector<float> vals;
    std::vector<uint32_t> word, pos, dst;
    std::vector<int32_t> chunk_t0;
    size_t ntiles() const { return word.size() / 64; }
};
static float blockval(uint32_t i, uint32_t j, int k)  // deterministic pseudo-random entry k of block (i, j), i <= j
{
    uint32_t h = i * 2654435761u ^ (j + 0x9e3779b9u) * 2246822519u ^ (uint32_t)k * 3266489917u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(h & 0xffff) / 65536.f - 0.5f;
}
static void block_of(uint32_t i, uint32_t j, float* b)  // row-major 3x3 of block (i, j) of a symmetric matrix
{
    if (i < j) for (int k = 0; k < 9; k++) b[k] = blockval(i, j, k);
    else if (i > j) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) b[3 * r + c] = blockval(j, i, 3 * c + r);
    else for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) b[3 * r + c] = (r == c ? 8.f : 0.f) + blockval(i, i, r <= c ? 3 * r + c : 3 * c + r);
}
static void put_vals(HostPart& P, size_t at, const float* b)
{
    float* tv = P.vals.data() + (at >> 6) * 576;
    const int l = (int)(at & 63);
    // tile layout of the engine: float4 a = (b00 b01 b02 b10) at [l], float4 b = (b11 b12 b20 b21) at [64 + l], cc = b22 at [512 + l]
    for (int k = 0; k < 4; k++) tv[4 * l + k] = b[k];
    for (int k = 0; k < 4; k++) tv[256 + 4 * l + k] = b[4 + k];
    tv[512 + l] = b[8];
}

int main(int argc, char** argv)
{
    const int nx = argc > 1 ? std::atoi(argv[1]) : 56, ny = argc > 2 ? std::atoi(argv[2]) : 56, nz = argc > 3 ? std::atoi(argv[3]) : 55;
    const int chunk_tiles = argc > 4 ? std::atoi(argv[4]) : 4, reps = argc > 5 ? std::atoi(argv[5]) : 200;
    const int64_t n = (int64_t)nx * ny * nz;
    // rows in Morton order
    std::vector<uint32_t> order((size_t)n), rank_of((size_t)n);
    std::iota(order.begin(), order.end(), 0u);
    auto coord = [&](uint32_t g, int& ix, int& iy, int& iz) { ix = g % nx; iy = (g / nx) % ny; iz = g / (nx * ny); };
    {
        std::vector<uint64_t> key((size_t)n);
        for (uint32_t g = 0; g < n; g++) {
            int ix, iy, iz;
            coord(g, ix, iy, iz);
            key[g] = morton3(ix, iy, iz);
        }
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        for (uint32_t r = 0; r < n; r++) rank_of[order[r]] = r;
    }
    static const int NB[14][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 1, 0}, {-1, -1, 0}, {0, 1, 1}, {0, -1, -1}, {1, 0, 1}, {-1, 0, -1}, {1, 1, 1}, {-1, -1, -1}};
    std::vector<std::vector<uint32_t>> cols((size_t)n);
    int64_t nnzb = 0;
    for (uint32_t r = 0; r < n; r++) {
        int ix, iy, iz;
        coord(order[r], ix, iy, iz);
        auto& c = cols[r];
        c.push_back(r);
        for (auto& d : NB) {
            const int jx = ix + d[0], jy = iy + d[1], jz = iz + d[2];
            if (jx < 0 || jy < 0 || jz < 0 || jx >= nx || jy >= ny || jz >= nz) continue;
            c.push_back(rank_of[(size_t)jx + (size_t)nx * (jy + (size_t)ny * jz)]);
        }
        std::sort(c.begin(), c.end());
        nnzb += (int64_t)c.size();
    }
    // ---- storages: rows cut into chunks of whole rows with at most chunk_tiles * 64 stored entries (the same rows for all streams of a scheme)
    auto build = [&](bool sym, HostPart& U, HostPart& Lw) {
        std::vector<size_t> pos_of_upper;  // sym: position of stored block (i, j), looked up by the gather stream: (row i -> positions in column order)
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> upper_pos(sym ? (size_t)n : 0);
        const size_t cap = (size_t)chunk_tiles * 64;
        std::vector<uint32_t> chunk_row0{0};
        {
            size_t in_chunk = 0;
            for (uint32_t r = 0; r < n; r++) {
                size_t cnt = 0;
                for (uint32_t c : cols[r]) cnt += (!sym || c >= r) ? 1 : 0;
                if (in_chunk + cnt > cap && in_chunk > 0) {
                    chunk_row0.push_back(r);
                    in_chunk = 0;
                }
                in_chunk += cnt;
            }
            chunk_row0.push_back((uint32_t)n);
        }
        const size_t nch = chunk_row0.size() - 1;
        auto emit = [&](HostPart& P, bool gather) {
            P.chunk_t0.assign(1, 0);
            for (size_t ch = 0; ch < nch; ch++) {
                size_t at = P.word.size();
                for (uint32_t r = chunk_row0[ch]; r < chunk_row0[ch + 1]; r++) {
                    std::vector<uint32_t> mine;
                    for (uint32_t c : cols[r])
                        if (gather ? c < r : (!sym || c >= r)) mine.push_back(c);
                    for (size_t k = 0; k < mine.size(); k++) {
                        const uint32_t c = mine[k];
                        const bool tail = k + 1 == mine.size();
                        P.word.push_back(c | (tail ? 0x80000000u : 0u));
                        P.dst.push_back(r);
                        if (!gather) {
                            if ((P.word.size() - 1) % 64 == 0) P.vals.resize(P.vals.size() + 576, 0.f);
                            float b[9];
                            block_of(r, c, b);
                            put_vals(P, P.word.size() - 1, b);
                            if (sym) upper_pos[r].push_back({c, (uint32_t)(P.word.size() - 1)});
                        } else {
                            // block (r, c), c < r = transpose of stored (c, r)
                            const auto& lst = upper_pos[c];
                            auto it = std::lower_bound(lst.begin(), lst.end(), std::make_pair(r, 0u));
                            P.pos.push_back(it->second);
                        }
                    }
                }
                (void)at;
                while (P.word.size() % 64) {  // chunks start on tile boundaries
                    P.word.push_back(PAD);
                    P.dst.push_back(0);
                    if (gather) P.pos.push_back(0);
                }
                if (!gather && P.vals.size() < P.word.size() / 64 * 576) P.vals.resize(P.word.size() / 64 * 576, 0.f);
                P.chunk_t0.push_back((int32_t)(P.word.size() / 64));
            }
        };
        emit(U, false);
        if (sym) emit(Lw, true);
        return nch;
    };
    HostPart F, Fdummy, U, Lw;
    const size_t nch_full = build(false, F, Fdummy);
    const size_t nch_sym = build(true, U, Lw);
    std::printf("grid %d x %d x %d: %lld block rows, %lld blocks (%.2f per row)\n", nx, ny, nz, (long long)n, (long long)nnzb, (double)nnzb / n);
    const double bytes_full = (double)F.ntiles() * (2304 + 256 + 256) + 48.0 * n;  // values + column words + destination rows; x gathered once per row on average + y
    const double bytes_sym = (double)U.ntiles() * (2304 + 256 + 256) + (double)Lw.ntiles() * (256 * 3) + 48.0 * n;
    const double bytes_engine = (double)nnzb * 40 + 8.0 * (n + 1) + 48.0 * n;  // the engine's accounting for its own format
    std::printf("full storage: %zu tiles in %zu chunks, %.1f MB | symmetric half: %zu stored + %zu gather tiles in %zu chunks, %.1f MB | engine's algorithmic bytes %.1f MB\n", F.ntiles(),
                nch_full, bytes_full / 1e6, U.ntiles(), Lw.ntiles(), nch_sym, bytes_sym / 1e6, bytes_engine / 1e6);

    // ---- device
    auto up = [&](const void* h, size_t bytes, void** d) -> hipError_t {
        hipError_t e = hipMalloc(d, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    auto dev_part = [&](const HostPart& H, Part& P) -> hipError_t {
        hipError_t e;
        void *v = nullptr, *w = nullptr, *p = nullptr, *d = nullptr, *c = nullptr;
        if ((e = up(H.vals.data(), H.vals.size() * 4, &v)) != hipSuccess) return e;
        if ((e = up(H.word.data(), H.word.size() * 4, &w)) != hipSuccess) return e;
        if ((e = up(H.pos.data(), H.pos.size() * 4, &p)) != hipSuccess) return e;
        if ((e = up(H.dst.data(), H.dst.size() * 4, &d)) != hipSuccess) return e;
        if ((e = up(H.chunk_t0.data(), H.chunk_t0.size() * 4, &c)) != hipSuccess) return e;
        P = Part{(const float*)v, (const uint32_t*)w, (const uint32_t*)p, (const uint32_t*)d, (const int32_t*)c};
        return hipSuccess;
    };
    Part dF{}, dU{}, dL{}, dNone{};
    CK(dev_part(F, dF));
    CK(dev_part(U, dU));
    CK(dev_part(Lw, dL));
    std::vector<double> hx(3 * (size_t)n);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = std::sin(0.37 * (double)i) + 0.25;
    double *x, *y1, *y2, *out;
    CK(hipMalloc(&x, hx.size() * 8));
    CK(hipMalloc(&y1, hx.size() * 8));
    CK(hipMalloc(&y2, hx.size() * 8));
    CK(hipMalloc(&out, 64));
    CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(y1, 0, hx.size() * 8));
    CK(hipMemset(y2, 0, hx.size() * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto grid_for = [&](size_t nch) { return (int)std::max<size_t>(std::min<size_t>(((nch + 3) / 4 + 7) / 8 * 8, 2048), 8); };
    const int g_full = grid_for(nch_full), g_sym = grid_for(nch_sym);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& launch, double& us) -> hipError_t {
        for (int w = 0; w < 10; w++) launch();
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1, s);
        e = hipEventSynchronize(e1);
        if (e != hipSuccess) return e;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us = 1e3 * ms / reps;
        return hipGetLastError();
    };
    double us_full = 0, us_sym = 0, us_stream_full = 0, us_stream_sym = 0;
    CK(time_it([&] { hipLaunchKernelGGL(k_spmv<false>, dim3(g_full), dim3(BLOCK), 0, s, dF, dNone, (int64_t)nch_full, (const double*)x, y1); }, us_full));
    CK(time_it([&] { hipLaunchKernelGGL(k_spmv<true>, dim3(g_sym), dim3(BLOCK), 0, s, dU, dL, (int64_t)nch_sym, (const double*)x, y2); }, us_sym));
    CK(time_it([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(BLOCK), 0, s, (const float4*)dF.vals, F.vals.size() / 4, out); }, us_stream_full));
    CK(time_it([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(BLOCK), 0, s, (const float4*)dU.vals, U.vals.size() / 4, out); }, us_stream_sym));
    // ---- check: both against a host product in double, and against each other
    std::vector<double> h1(hx.size()), h2(hx.size());
    CK(hipMemcpy(h1.data(), y1, h1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), y2, h2.size() * 8, hipMemcpyDeviceToHost));
    double err_full = 0, err_sym = 0, diff = 0, ymax = 0;
    const int64_t n_check = std::min<int64_t>(n, 20000);
    for (int64_t q = 0; q < n_check; q++) {
        const uint32_t r = (uint32_t)((q * 7919) % n);
        double ref[3] = {0, 0, 0};
        for (uint32_t c : cols[r]) {
            float b[9];
            block_of(r, c, b);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) ref[i] += (double)b[3 * i + j] * hx[3 * (size_t)c + j];
        }
        for (int i = 0; i < 3; i++) {
            ymax = std::max(ymax, std::fabs(ref[i]));
            err_full = std::max(err_full, std::fabs(h1[3 * (size_t)r + i] - ref[i]));
            err_sym = std::max(err_sym, std::fabs(h2[3 * (size_t)r + i] - ref[i]));
        }
    }
    for (size_t i = 0; i < h1.size(); i++) diff = std::max(diff, std::fabs(h1[i] - h2[i]));
    std::printf("check (%lld rows against a host product): full %.2e, symmetric %.2e of max |y| = %.3f; full vs symmetric on all rows %.2e\n", (long long)n_check, err_full / ymax,
                err_sym / ymax, ymax, diff / ymax);
    std::printf("full storage      : %8.2f us per launch  (%6.0f GB/s of its own %.1f MB; %6.0f GB/s of the engine's %.1f MB)   value stream alone %7.2f us\n", us_full,
                bytes_full / us_full / 1e3, bytes_full / 1e6, bytes_engine / us_full / 1e3, bytes_engine / 1e6, us_stream_full);
    std::printf("symmetric half    : %8.2f us per launch  (%6.0f GB/s of its own %.1f MB; %6.0f GB/s of the engine's %.1f MB)   value stream alone %7.2f us\n", us_sym,
                bytes_sym / us_sym / 1e3, bytes_sym / 1e6, bytes_engine / us_sym / 1e3, bytes_engine / 1e6, us_stream_sym);
    std::printf("symmetric / full  : %.3f\n", us_sym / us_full);
    return (err_full / ymax < 1e-12 && err_sym / ymax < 1e-12) ? 0 : 2;
}
