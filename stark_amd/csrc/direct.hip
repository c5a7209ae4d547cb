// direct.hip — the reference's DirectLLT alternative to the block-Jacobi PCG (symx/src/solver/NewtonsMethod.cpp:395-418: the float BSR
// Hessian as double triplets, Eigen::SimplicialLLT, du = A^-1 (-grad); any non-positive pivot = "solve failed").
//
// The reference uses it for small stiff rigid-body problems (tests/rb_constraints.cpp:40). Two paths:
//  * up to MAX_DIRECT_DOFS unknowns: a dense Cholesky in ONE workgroup (right-looking, the trailing update spread over its threads): the
//    rigid-body test systems have a dozen unknowns, a library call would be all latency;
//  * beyond: a BLOCK-TRIDIAGONAL Cholesky. The block rows are renumbered by reverse Cuthill-McKee (host, once per sparsity pattern); with
//    blocks of at least the bandwidth the matrix is block tridiagonal, A = tridiag(S_{i-1}, D_i, S_i^T), and
//        L_ii = chol(D_i);  L_{i+1,i} = S_i L_ii^-T;  D_{i+1} -= L_{i+1,i} L_{i+1,i}^T
//    runs on three small kernels here (k_gemm_nt_sub: LDS-tiled C -= A B^T; k_chol_tile / k_trsm_tile: a 64 x 64 diagonal tile in one
//    workgroup): every block column [D_i; S_i] is one tall column-major panel factored LEFT-looking in 64-column steps, so all the heavy
//    work is the tiled product with a long inner dimension. (A first version called rocSOLVER potrf / rocBLAS trsm, syrk: 0.23 s per
//    factorisation of configs[1], the same as these kernels, but the first call paid the library's cold start: 2.5 s on one test box,
//    227 s on another.) The fill-in of SimplicialLLT lives inside the band, so storing the band densely costs memory (2 m^2 doubles per
//    block of m unknowns), not correctness; the size is checked against MISTARK_DIRECT_MAX_GB (default 64).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "engine.hpp"

namespace mistark {

constexpr int MAX_DIRECT_DOFS = 3072;
constexpr int DT = 1024;

namespace {
// dense (column-major, full storage) += the 3x3 blocks of one matrix part
__global__ __launch_bounds__(256) void k_dense_add(const float* __restrict__ vals, const uint32_t* __restrict__ colw, const uint32_t* __restrict__ slot_row,
                                                   const uint32_t* __restrict__ store_slot, int64_t nnzb, int n, double* __restrict__ A)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nnzb * 9) return;
    const int64_t s = t / 9;
    const int comp = (int)(t - s * 9);
    const size_t pos = store_slot ? (size_t)store_slot[s] : (size_t)s;  // (static part: chunk-aligned storage)
    const size_t base = (pos >> 6) * 576;
    const size_t lane = pos & 63;
    const size_t idx = comp < 4 ? base + lane * 4 + comp : (comp < 8 ? base + 256 + lane * 4 + (comp - 4) : base + 512 + lane);
    const int row = 3 * (int)slot_row[s] + comp / 3, col = 3 * (int)(colw[s] & 0x7fffffffu) + comp % 3;
    A[(size_t)col * n + row] += (double)vals[idx];  // (row, col) pairs are unique within a part
}
__global__ __launch_bounds__(DT) void k_cholesky_solve(double* __restrict__ A, int n, const double* __restrict__ rhs, double* __restrict__ x, int* __restrict__ status)
{
    __shared__ int failed;
    const int tid = threadIdx.x;
    if (tid == 0) failed = 0;
    __syncthreads();
    for (int k = 0; k < n; k++) {
        double* colk = A + (size_t)k * n;
        if (tid == 0) {
            const double d = colk[k];
            if (!(d > 0.0)) failed = 1;  // SimplicialLLT: numerical issue -> info() != Success
            else colk[k] = sqrt(d);
        }
        __syncthreads();
        if (failed) {
            if (tid == 0) *status = 1;
            return;
        }
        const double inv = 1.0 / colk[k];
        for (int i = k + 1 + tid; i < n; i += DT) colk[i] *= inv;
        __syncthreads();
        const int m = n - k - 1;
        // trailing lower triangle: A[i][j] -= L[i][k] L[j][k], i >= j > k; one column per group of threads keeps the accesses coalesced
        for (int64_t e = tid; e < (int64_t)m * m; e += DT) {
            const int j = (int)(e / m), i = (int)(e - (int64_t)j * m);
            if (i >= j) A[(size_t)(k + 1 + j) * n + (k + 1 + i)] -= colk[k + 1 + i] * colk[k + 1 + j];
        }
        __syncthreads();
    }
    // L y = rhs (y in x), then L^T x = y
    for (int i = tid; i < n; i += DT) x[i] = rhs[i];
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const double* colk = A + (size_t)k * n;
        if (tid == 0) x[k] /= colk[k];
        __syncthreads();
        const double yk = x[k];
        for (int i = k + 1 + tid; i < n; i += DT) x[i] -= colk[i] * yk;
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; k--) {
        const double* colk = A + (size_t)k * n;
        // x[k] = (y[k] - sum_{i > k} L[i][k] x[i]) / L[k][k]
        __shared__ double part[DT / 64];
        double s = 0.0;
        for (int i = k + 1 + tid; i < n; i += DT) s += colk[i] * x[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((tid & 63) == 0) part[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < DT / 64; w++) t += part[w];
            x[k] = (x[k] - t) / colk[k];
        }
        __syncthreads();
    }
    if (tid == 0) *status = 0;
}
// ---- dense kernels of the block-tridiagonal Cholesky (column-major, leading dimension ld) ------------------------------------------------
constexpr int TS = 64;   // tile
constexpr int KC = 16;   // inner chunk
// C (M x N) -= A (M x K) B (N x K)^T. One workgroup per 64 x 64 tile of C, 256 threads x (4 x 4) outputs, A and B chunks of 16 columns staged
// through LDS. lower: C is a symmetric target of which only the lower triangle is needed (tiles strictly above the diagonal are skipped).
__global__ __launch_bounds__(256) void k_gemm_nt_sub(int M, int N, int K, const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, double* __restrict__ C,
                                                     int64_t ldc, int lower)
{
    const int ti = blockIdx.x * TS, tj = blockIdx.y * TS;
    if (lower && tj > ti + TS - 1) return;
    __shared__ double As[KC][TS + 1], Bs[KC][TS + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int t = threadIdx.x; t < KC * TS; t += 256) {
            const int kk = t / TS, r = t - kk * TS;
            As[kk][r] = (ti + r < M && k0 + kk < K) ? A[(int64_t)(ti + r) + (int64_t)(k0 + kk) * lda] : 0.0;
            Bs[kk][r] = (tj + r < N && k0 + kk < K) ? B[(int64_t)(tj + r) + (int64_t)(k0 + kk) * ldb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = As[kk][tx + 16 * u];
                b[u] = Bs[kk][ty + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[u][v] += a[u] * b[v];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = ti + tx + 16 * u, j = tj + ty + 16 * v;
            if (i < M && j < N) C[(int64_t)i + (int64_t)j * ldc] -= acc[u][v];
        }
}
// Cholesky of an n x n tile (n <= 64), lower, in place; *info = 1 on a non-positive pivot
__global__ __launch_bounds__(256) void k_chol_tile(int n, double* __restrict__ A, int64_t lda, int* __restrict__ info)
{
    __shared__ double T[TS][TS + 1];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    for (int t = threadIdx.x; t < n * n; t += 256) {
        const int j = t / n, i = t - j * n;
        T[i][j] = A[(int64_t)i + (int64_t)j * lda];
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        if (threadIdx.x == 0) {
            const double d = T[k][k];
            if (!(d > 0.0)) bad = 1;
            else T[k][k] = sqrt(d);
        }
        __syncthreads();
        if (bad) {
            if (threadIdx.x == 0) *info = 1;
            return;
        }
        const double inv = 1.0 / T[k][k];
        for (int i = k + 1 + (int)threadIdx.x; i < n; i += 256) T[i][k] *= inv;
        __syncthreads();
        const int m = n - k - 1;
        for (int t = threadIdx.x; t < m * m; t += 256) {
            const int j = t / m, i = t - j * m;
            if (i >= j) T[k + 1 + i][k + 1 + j] -= T[k + 1 + i][k] * T[k + 1 + j][k];
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < n * n; t += 256) {
        const int j = t / n, i = t - j * n;
        if (i >= j) A[(int64_t)i + (int64_t)j * lda] = T[i][j];
    }
}
// X (M x n) <- X L^-T with L an n x n lower tile (n <= 64): one thread per row of X, 64 rows per workgroup, the rows and L in LDS
__global__ __launch_bounds__(64) void k_trsm_tile(int M, int n, const double* __restrict__ L, int64_t ldl, double* __restrict__ X, int64_t ldx)
{
    __shared__ double T[TS][TS + 1], Xs[TS][TS + 1];
    const int r0 = blockIdx.x * TS;
    for (int t = threadIdx.x; t < n * n; t += 64) {
        const int j = t / n, i = t - j * n;
        T[i][j] = L[(int64_t)i + (int64_t)j * ldl];
    }
    for (int t = threadIdx.x; t < n * TS; t += 64) {
        const int j = t / TS, i = t - j * TS;
        Xs[i][j] = r0 + i < M ? X[(int64_t)(r0 + i) + (int64_t)j * ldx] : 0.0;
    }
    __syncthreads();
    const int r = threadIdx.x;
    for (int j = 0; j < n; j++) {
        double v = Xs[r][j];
        for (int p = 0; p < j; p++) v -= Xs[r][p] * T[j][p];
        Xs[r][j] = v / T[j][j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n * TS; t += 64) {
        const int j = t / TS, i = t - j * TS;
        if (r0 + i < M) X[(int64_t)(r0 + i) + (int64_t)j * ldx] = Xs[i][j];
    }
}
// triangular solves of one 64-column step. Forward: y_k <- L_kk^-1 y_k (k_trsv_tile), then the rows below: y_r -= L[r, k] y_k (k_gemv_sub).
// Backward: y_k -= L[rows below, k]^T y_below (k_gemv_t_sub), then y_k <- L_kk^-T y_k.
__global__ __launch_bounds__(64) void k_trsv_tile(int n, const double* __restrict__ L, int64_t ldl, double* __restrict__ y, int transposed)
{
    __shared__ double T[TS][TS + 1], v[TS];
    for (int t = threadIdx.x; t < n * n; t += 64) {
        const int j = t / n, i = t - j * n;
        T[i][j] = L[(int64_t)i + (int64_t)j * ldl];
    }
    if ((int)threadIdx.x < n) v[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    if (!transposed) {
        for (int k = 0; k < n; k++) {
            if ((int)threadIdx.x == k) v[k] /= T[k][k];
            __syncthreads();
            if ((int)threadIdx.x > k && (int)threadIdx.x < n) v[threadIdx.x] -= T[threadIdx.x][k] * v[k];
            __syncthreads();
        }
    } else {
        for (int k = n - 1; k >= 0; k--) {
            if ((int)threadIdx.x == k) v[k] /= T[k][k];
            __syncthreads();
            if ((int)threadIdx.x < k) v[threadIdx.x] -= T[k][threadIdx.x] * v[k];
            __syncthreads();
        }
    }
    if ((int)threadIdx.x < n) y[threadIdx.x] = v[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_gemv_sub(int M, int n, const double* __restrict__ A, int64_t lda, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double xs[TS];
    if ((int)threadIdx.x < n) xs[threadIdx.x] = x[threadIdx.x];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    double s = 0.0;
    for (int j = 0; j < n; j++) s += A[(int64_t)r + (int64_t)j * lda] * xs[j];
    y[r] -= s;
}
// y_k[j] -= sum_r A[r, j] x[r]: one wavefront per column j
__global__ __launch_bounds__(256) void k_gemv_t_sub(int M, int n, const double* __restrict__ A, int64_t lda, const double* __restrict__ x, double* __restrict__ y)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    double s = 0.0;
    for (int r = lane; r < M; r += 64) s += A[(int64_t)r + (int64_t)j * lda] * x[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[j] -= s;
}
// permuted block-tridiagonal storage += the 3x3 blocks of one matrix part (lower triangle of the permuted matrix only)
__global__ __launch_bounds__(256) void k_blocktri_add(const float* __restrict__ vals, const uint32_t* __restrict__ colw, const uint32_t* __restrict__ slot_row,
                                                      const uint32_t* __restrict__ store_slot, int64_t nnzb, const int32_t* __restrict__ perm, int mb, int64_t n,
                                                      double* __restrict__ D, double* __restrict__ S, int* __restrict__ status)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nnzb * 9) return;
    const int64_t s = t / 9;
    const int comp = (int)(t - s * 9);
    const size_t pos = store_slot ? (size_t)store_slot[s] : (size_t)s;
    const size_t base = (pos >> 6) * 576;
    const size_t lane = pos & 63;
    const size_t idx = comp < 4 ? base + lane * 4 + comp : (comp < 8 ? base + 256 + lane * 4 + (comp - 4) : base + 512 + lane);
    const int64_t row = 3 * (int64_t)perm[slot_row[s]] + comp / 3, col = 3 * (int64_t)perm[colw[s] & 0x7fffffffu] + comp % 3;
    if (row < col) return;  // (the upper triangle mirrors it)
    const int64_t m = 3 * (int64_t)mb;
    const int64_t bi = row / m, bj = col / m;
    const int64_t li = row - bi * m, lj = col - bj * m;
    // block column bj is one tall column-major panel of 2m x m (leading dimension 2m): rows [0, m) = D_bj, rows [m, 2m) = S_bj = A_{bj+1, bj}
    if (bi == bj) atomicAdd(&D[(size_t)bj * 2 * m * m + (size_t)lj * 2 * m + li], (double)vals[idx]);
    else if (bi == bj + 1) atomicAdd(&D[(size_t)bj * 2 * m * m + (size_t)lj * 2 * m + m + li], (double)vals[idx]);
    else *status = 2;  // outside the band: the ordering is not the one the blocks were sized for
    (void)n;
    (void)S;
}
__global__ __launch_bounds__(256) void k_permute(const double* __restrict__ src, const int32_t* __restrict__ perm, int64_t nbr, bool forward, double* __restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * nbr) return;
    const int64_t r = t / 3, c = t - 3 * r;
    if (forward) dst[3 * (int64_t)perm[r] + c] = src[t];
    else dst[t] = src[3 * (int64_t)perm[r] + c];
}

// reverse Cuthill-McKee of the block rows from the (row, column) pairs of the assembled pattern; returns the block bandwidth
int64_t rcm_order(int64_t nbr, const std::vector<uint32_t>& rows, const std::vector<uint32_t>& cols, std::vector<int32_t>& perm)
{
    std::vector<int64_t> start((size_t)nbr + 1, 0);
    for (size_t k = 0; k < rows.size(); k++)
        if (rows[k] != cols[k]) start[(size_t)rows[k] + 1]++;
    for (int64_t r = 0; r < nbr; r++) start[(size_t)r + 1] += start[(size_t)r];
    std::vector<uint32_t> adj((size_t)start[(size_t)nbr]);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (size_t k = 0; k < rows.size(); k++)
            if (rows[k] != cols[k]) adj[(size_t)fill[rows[k]]++] = cols[k];
    }
    auto deg = [&](int64_t r) { return start[(size_t)r + 1] - start[(size_t)r]; };
    std::vector<int32_t> level((size_t)nbr, -1);
    std::vector<int64_t> order;
    order.reserve((size_t)nbr);
    auto bfs = [&](int64_t root, std::vector<int64_t>& out) {
        const size_t first = out.size();
        out.push_back(root);
        level[(size_t)root] = 0;
        std::vector<uint32_t> nb;
        for (size_t h = first; h < out.size(); h++) {
            const int64_t u = out[h];
            nb.clear();
            for (int64_t j = start[(size_t)u]; j < start[(size_t)u + 1]; j++)
                if (level[adj[(size_t)j]] < 0) {
                    level[adj[(size_t)j]] = level[(size_t)u] + 1;
                    nb.push_back(adj[(size_t)j]);
                }
            std::sort(nb.begin(), nb.end(), [&](uint32_t a, uint32_t b) { return deg(a) < deg(b) || (deg(a) == deg(b) && a < b); });  // Cuthill-McKee: by degree
            for (uint32_t v : nb) out.push_back(v);
        }
        return out.back();
    };
    for (int64_t r0 = 0; r0 < nbr; r0++) {
        if (level[(size_t)r0] >= 0) continue;
        std::vector<int64_t> tmp;
        int64_t far = bfs(r0, tmp);
        for (int64_t v : tmp) level[(size_t)v] = -1;
        tmp.clear();
        far = bfs(far, tmp);
        for (int64_t v : tmp) level[(size_t)v] = -1;
        bfs(far, order);
    }
    perm.assign((size_t)nbr, 0);
    for (int64_t i = 0; i < nbr; i++) perm[(size_t)order[(size_t)(nbr - 1 - i)]] = (int32_t)i;  // reversed
    int64_t bw = 0;
    for (size_t k = 0; k < rows.size(); k++) bw = std::max<int64_t>(bw, std::abs((int64_t)perm[rows[k]] - (int64_t)perm[cols[k]]));
    return bw;
}

bool direct_llt_blocktri(Context& c, const double* rhs_dev, double* x_dev)
{
    const int64_t nbr = c.nbr, n = c.ndofs;
    static const bool trace = std::getenv("MISTARK_LLT_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(c.stream);
        const double t = now();
        std::fprintf(stderr, "[llt] %-28s %.3f s\n", what, t - t_prev);
        t_prev = t;
    };
    // ---- ordering and block size: once per sparsity pattern
    if (c.llt_pattern_version != c.pattern_version) {
        std::vector<uint32_t> rows, cols;
        for (int part = 0; part < 2; part++) {
            const BsrPart& m = c.part[part];
            if (m.nnzb == 0) continue;
            std::vector<uint32_t> cw((size_t)m.nnzb), rw((size_t)m.nnzb);
            MS_CHECK(hipMemcpyAsync(cw.data(), m.colw.p, cw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
            MS_CHECK(hipMemcpyAsync(rw.data(), m.slot_row.p, rw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
            MS_CHECK(hipStreamSynchronize(c.stream));
            for (int64_t s = 0; s < m.nnzb; s++) {
                rows.push_back(rw[(size_t)s]);
                cols.push_back(cw[(size_t)s] & 0x7fffffffu);
            }
        }
        std::vector<int32_t> perm;
        const int64_t bw = rcm_order(nbr, rows, cols, perm);
        c.llt_mb = (int)std::min<int64_t>(nbr, std::max<int64_t>(bw, 256));  // block rows per dense block: at least the bandwidth
        c.llt_perm.ensure((size_t)nbr);
        MS_CHECK(hipMemcpyAsync(c.llt_perm.p, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
        c.llt_pattern_version = c.pattern_version;
        if (trace) std::fprintf(stderr, "[llt] block rows %lld, bandwidth %lld, block size %d\n", (long long)nbr, (long long)bw, c.llt_mb);
        lap("ordering");
    }
    const int64_t mb = c.llt_mb, m = 3 * mb;
    const int64_t N = (nbr + mb - 1) / mb;
    const double gb = (double)(2 * N) * (double)m * (double)m * 8.0 / 1e9;  // N tall panels of 2m x m
    const char* cap_env = std::getenv("MISTARK_DIRECT_MAX_GB");
    const double cap = cap_env ? std::atof(cap_env) : 64.0;
    if (gb > cap)
        throw Error("DirectLLT: the band of this system (" + std::to_string(n) + " unknowns, half bandwidth " + std::to_string(m) + ") needs " + std::to_string((int)gb) +
                    " GB as dense blocks (limit MISTARK_DIRECT_MAX_GB = " + std::to_string((int)cap) + "); use the block-Jacobi PCG");
    const int64_t ld = 2 * m;
    const size_t panel = (size_t)ld * (size_t)m;  // doubles per block column
    c.llt_D.ensure((size_t)N * panel);
    c.llt_y.ensure((size_t)n);
    c.llt_info.ensure(2);
    MS_CHECK(hipMemsetAsync(c.llt_D.p, 0, (size_t)N * panel * sizeof(double), c.stream));
    MS_CHECK(hipMemsetAsync(c.llt_info.p, 0, 2 * sizeof(int), c.stream));
    int* status = c.llt_info.p;
    for (int part = 0; part < 2; part++) {
        const BsrPart& mp = c.part[part];
        if (mp.nnzb == 0) continue;
        hipLaunchKernelGGL(k_blocktri_add, dim3((unsigned)((mp.nnzb * 9 + 255) / 256)), dim3(256), 0, c.stream, mp.vals.p, mp.colw.p, mp.slot_row.p,
                           (part == 0 && mp.n_chunks_static > 0) ? (const uint32_t*)mp.store_slot.p : (const uint32_t*)nullptr, mp.nnzb, (const int32_t*)c.llt_perm.p, (int)mb, n,
                           c.llt_D.p, (double*)nullptr, status);
    }
    lap("fill blocks");
    auto msize = [&](int64_t i) { return (int)std::min<int64_t>(m, n - i * m); };
    auto tiles = [](int x) { return (unsigned)((x + TS - 1) / TS); };
    for (int64_t i = 0; i < N; i++) {
        double* T = c.llt_D.p + (size_t)i * panel;       // [D_i; S_i], ld = 2m
        const int mi = msize(i), mnext = i + 1 < N ? msize(i + 1) : 0;
        // the rows of the tall panel: D_i and, below it, S_i (contiguous: only the LAST block column can have a short D, and it has no S)
        const int R = mnext > 0 ? (int)m + mnext : mi;
        // right-looking over 256-column panels (the trailing update is one product over thousands of tiles), left-looking over 64-column steps
        // inside a panel
        constexpr int PW = 256;
        for (int pb = 0; pb < mi; pb += PW) {
            const int pw = std::min(PW, mi - pb);
            for (int kb = pb; kb < pb + pw; kb += TS) {
                const int nb = std::min(TS, pb + pw - kb);
                double* Ckk = T + kb + (size_t)kb * ld;       // row kb, column kb
                const double* Pk = T + kb + (size_t)pb * ld;  // row kb, first column of the panel
                if (kb > pb) hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(R - kb), 1), dim3(256), 0, c.stream, R - kb, nb, kb - pb, Pk, ld, Pk, ld, Ckk, ld, 0);
                hipLaunchKernelGGL(k_chol_tile, dim3(1), dim3(256), 0, c.stream, nb, Ckk, ld, status + 1);
                if (R - kb - nb > 0) hipLaunchKernelGGL(k_trsm_tile, dim3(tiles(R - kb - nb)), dim3(64), 0, c.stream, R - kb - nb, nb, (const double*)Ckk, ld, Ckk + nb, ld);
            }
            const int e = pb + pw, nc = mi - e;
            if (nc > 0) {
                const double* Pe = T + e + (size_t)pb * ld;
                hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(R - e), tiles(nc)), dim3(256), 0, c.stream, R - e, nc, pw, Pe, ld, Pe, ld, T + e + (size_t)e * ld, ld, 1);
            }
        }
        // D_{i+1} -= S_i S_i^T (lower triangle)
        if (mnext > 0)
            hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(mnext), tiles(mnext)), dim3(256), 0, c.stream, mnext, mnext, mi, (const double*)(T + m), ld, (const double*)(T + m), ld, T + panel, ld, 1);
    }
    lap("factorisation");
    int h[2] = {1, 1};
    fetch(c, h, status, 2 * sizeof(int));
    if (h[0] == 2) throw Error("DirectLLT: internal: a block fell outside the band");
    if (h[1] != 0) return false;  // a non-positive pivot: SimplicialLLT's info() != Success
    // ---- L y = P b, L^T z = y, x = P^T z
    double* y = c.llt_y.p;
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((3 * nbr + 255) / 256)), dim3(256), 0, c.stream, rhs_dev, (const int32_t*)c.llt_perm.p, nbr, true, y);
    for (int64_t i = 0; i < N; i++) {
        const double* T = c.llt_D.p + (size_t)i * panel;
        const int mi = msize(i), mnext = i + 1 < N ? msize(i + 1) : 0;
        double* yi = y + i * m;
        for (int kb = 0; kb < mi; kb += TS) {
            const int nb = std::min(TS, mi - kb);
            const double* Ckk = T + kb + (size_t)kb * ld;
            hipLaunchKernelGGL(k_trsv_tile, dim3(1), dim3(64), 0, c.stream, nb, Ckk, ld, yi + kb, 0);
            const int below = mi - kb - nb;
            if (below > 0) hipLaunchKernelGGL(k_gemv_sub, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, c.stream, below, nb, Ckk + nb, ld, (const double*)(yi + kb), yi + kb + nb);
            if (mnext > 0) hipLaunchKernelGGL(k_gemv_sub, dim3((unsigned)((mnext + 255) / 256)), dim3(256), 0, c.stream, mnext, nb, T + m + (size_t)kb * ld, ld, (const double*)(yi + kb), yi + m);
        }
    }
    for (int64_t i = N - 1; i >= 0; i--) {
        const double* T = c.llt_D.p + (size_t)i * panel;
        const int mi = msize(i), mnext = i + 1 < N ? msize(i + 1) : 0;
        double* yi = y + i * m;
        for (int kb = (mi - 1) / TS * TS; kb >= 0; kb -= TS) {
            const int nb = std::min(TS, mi - kb);
            const double* Ckk = T + kb + (size_t)kb * ld;
            const int below = mi - kb - nb;
            if (below > 0) hipLaunchKernelGGL(k_gemv_t_sub, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c.stream, below, nb, Ckk + nb, ld, (const double*)(yi + kb + nb), yi + kb);
            if (mnext > 0) hipLaunchKernelGGL(k_gemv_t_sub, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c.stream, mnext, nb, T + m + (size_t)kb * ld, ld, (const double*)(yi + m), yi + kb);
            hipLaunchKernelGGL(k_trsv_tile, dim3(1), dim3(64), 0, c.stream, nb, Ckk, ld, yi + kb, 1);
        }
    }
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((3 * nbr + 255) / 256)), dim3(256), 0, c.stream, (const double*)y, (const int32_t*)c.llt_perm.p, nbr, false, x_dev);
    lap("triangular solves");
    return true;
}
}  // namespace

// du = A^-1 rhs with A = A_static + A_dynamic as assembled. Returns false when the factorisation meets a non-positive pivot.
static bool direct_llt_solver_numbering(Context& c, const double* rhs_dev, double* x_dev);
bool direct_llt(Context& c, const double* rhs_dev, double* x_dev)
{
    if (!c.have_matrix) throw Error("direct_llt: matrix not assembled");
    if (c.world > 1) throw Error("DirectLLT is a single-rank solver (a sharded context holds its own rows only); use the block-Jacobi PCG");
    if (!c.perm_active) return direct_llt_solver_numbering(c, rhs_dev, x_dev);
    // the matrix lives in solver numbering (Context::perm_active): right-hand side in, solution out (the PCG's vectors are free here)
    rows_to_solver(c, rhs_dev, c.p.p);
    const bool ok = direct_llt_solver_numbering(c, c.p.p, c.q.p);
    rows_from_solver(c, c.q.p, x_dev);
    return ok;
}
static bool direct_llt_solver_numbering(Context& c, const double* rhs_dev, double* x_dev)
{
    const int n = (int)c.ndofs;
    if (c.ndofs > MAX_DIRECT_DOFS) return direct_llt_blocktri(c, rhs_dev, x_dev);
    c.dense.ensure((size_t)n * n);
    c.counters.ensure(8);
    MS_CHECK(hipMemsetAsync(c.dense.p, 0, (size_t)n * n * sizeof(double), c.stream));
    for (int part = 0; part < 2; part++) {
        const BsrPart& m = c.part[part];
        if (m.nnzb == 0) continue;
        hipLaunchKernelGGL(k_dense_add, dim3((unsigned)((m.nnzb * 9 + 255) / 256)), dim3(256), 0, c.stream, m.vals.p, m.colw.p, m.slot_row.p, (part == 0 && m.n_chunks_static > 0) ? (const uint32_t*)m.store_slot.p : (const uint32_t*)nullptr, m.nnzb, n, c.dense.p);
    }
    int* status = reinterpret_cast<int*>(c.counters.p + 7);
    hipLaunchKernelGGL(k_cholesky_solve, dim3(1), dim3(DT), 0, c.stream, c.dense.p, n, rhs_dev, x_dev, status);
    int h = 1;
    fetch(c, &h, status, sizeof(int));
    return h == 0;
}

}  // namespace mistark
