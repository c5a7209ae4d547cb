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
//    block of m unknowns), not correctness;
//  * when the band costs more than 2 GB (a 3-D mesh: a whole cross-section wide): a MULTIFRONTAL Cholesky on a nested-dissection ordering
//    (second half of this file). configs[3]'s 517 044 unknowns: 7.1 GB of factor + 1.7 GB of update matrices, 1.9 s per factorisation, 0.9 s per
//    pair of triangular solves, residual 1.3e-15 (the band would need 390 GB). The size is checked against MISTARK_DIRECT_MAX_GB (default 64).
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

bool direct_llt_multifrontal(Context& c, const double* rhs_dev, double* x_dev, double cap_gb);
constexpr double MF_BAND_LIMIT_GB = 2.0;
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
    const char* cap_env0 = std::getenv("MISTARK_DIRECT_MAX_GB");
    const double cap0 = cap_env0 ? std::atof(cap_env0) : 64.0;
    if (c.llt_multifrontal > 0) return direct_llt_multifrontal(c, rhs_dev, x_dev, cap0);
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
    // a band that costs gigabytes (a 3-D mesh: a whole cross-section wide) goes to the multifrontal path (option llt_multifrontal: 1 = always,
    // -1 = never)
    if (c.llt_multifrontal == 0 && gb > std::min(cap, MF_BAND_LIMIT_GB)) return direct_llt_multifrontal(c, rhs_dev, x_dev, cap);
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

// ======================================================================================================================================================
// Beyond the band: a MULTIFRONTAL Cholesky on a nested-dissection ordering. The band of a 3-D mesh is a whole cross-section wide (configs[3]:
// 12 k unknowns, 50 GB of dense blocks); SimplicialLLT's fill is O(n^(4/3)). Here:
//  * ordering (host, once per sparsity pattern): recursive bisection of the block-row graph — by the median plane across the rows' positions
//    when the caller handed them over (the lower side's rows that touch the upper side are the separator), otherwise by breadth-first level
//    sets from a pseudo-peripheral row (the middle level is the separator: levels only touch their neighbours, so the two sides share no edge); rows of very high degree (a
//    rigid body every surface node is coupled to) leave the graph first and form the root. Parts of <= MF_LEAF rows are leaves.
//  * every tree node is a FRONT: its own rows S (eliminated there) and the later rows B its subtree couples to. The dense front
//    [F11; F21] (3(s+b) x 3s, column-major) receives the matrix entries of its columns and the update matrices of its children (extend-add
//    through precomputed relative indices), is factored as one tall panel with the kernels above (L11 = chol F11, L21 = F21 L11^-T), and hands
//    U = F22 - L21 L21^T (3b x 3b) to its parent. Panels stay (the factor), update matrices live in an arena with their lifetimes planned on
//    the host.
//  * solves walk the tree up (forward) and down (backward); the boundary rows of a front are gathered into / scattered from a dense work vector.
// Same answer as the band path to rounding (another elimination order); a non-positive pivot anywhere = "solve failed", as there.
constexpr int MF_LEAF = 192;  // block rows
struct MfFront
{
    int64_t off = 0;       // first new index of S
    int s = 0, b = 0;      // block rows eliminated here / boundary block rows
    int parent = -1;
    size_t rows_off = 0;   // B as new indices, in front_rows
    size_t panel_off = 0;  // doubles, in panels
    size_t u_off = 0;      // doubles, in the arena (b > 0)
    size_t rel_off = 0;    // position of each of its boundary rows in the PARENT's numbering (0 .. s_p + b_p), in rel
    std::vector<int> children;
};
struct Multifrontal
{
    std::vector<MfFront> fronts;  // post-order
    std::vector<int32_t> perm;    // block row -> new index
    DevBuf<int32_t> perm_dev, front_of_dev, front_rows_dev, rel_dev, fr_off_dev, fr_s_dev, fr_b_dev;
    DevBuf<int64_t> fr_rows_off_dev, fr_panel_off_dev;
    DevBuf<double> panels, arena, work;
    size_t panel_doubles = 0, arena_doubles = 0;
    int max_nf = 0;
};
namespace {
// F[(row of i in front f), (column j - off_f)] += entry; one thread per (block, component), lower triangle of the permuted matrix
__global__ __launch_bounds__(256) void k_mf_add(const float* __restrict__ vals, const uint32_t* __restrict__ colw, const uint32_t* __restrict__ slot_row,
                                                const uint32_t* __restrict__ store_slot, int64_t nnzb, const int32_t* __restrict__ perm, const int32_t* __restrict__ front_of,
                                                const int32_t* __restrict__ fr_off, const int32_t* __restrict__ fr_s, const int32_t* __restrict__ fr_b,
                                                const int64_t* __restrict__ fr_rows_off, const int64_t* __restrict__ fr_panel_off, const int32_t* __restrict__ front_rows,
                                                double* __restrict__ panels, int* __restrict__ status)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nnzb * 9) return;
    const int64_t sl = t / 9;
    const int comp = (int)(t - sl * 9);
    const size_t pos = store_slot ? (size_t)store_slot[sl] : (size_t)sl;
    const size_t base = (pos >> 6) * 576;
    const size_t lane = pos & 63;
    const size_t idx = comp < 4 ? base + lane * 4 + comp : (comp < 8 ? base + 256 + lane * 4 + (comp - 4) : base + 512 + lane);
    const int bi = perm[slot_row[sl]], bj = perm[colw[sl] & 0x7fffffffu];
    const int ci = comp / 3, cj = comp % 3;
    if (bi < bj || (bi == bj && ci < cj)) return;  // (the upper triangle mirrors it)
    const int f = front_of[bj];
    const int off = fr_off[f], sf = fr_s[f], bf = fr_b[f];
    int li;
    if (bi < off + sf) li = bi - off;
    else {  // position among the boundary rows (sorted)
        const int32_t* B = front_rows + fr_rows_off[f];
        int lo = 0, hi = bf;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (B[mid] < bi) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= bf || B[lo] != bi) {
            *status = 2;  // an entry outside the front's structure: the symbolic factorisation is wrong
            return;
        }
        li = sf + lo;
    }
    const int64_t ld = 3 * (int64_t)(sf + bf);
    atomicAdd(&panels[fr_panel_off[f] + (size_t)(3 * (bj - off) + cj) * ld + (size_t)(3 * li + ci)], (double)vals[idx]);
}
// parent front (P: panel, ld_p rows, 3 s_p columns; Up: 3 b_p square) += child's update matrix Uc (3 b_c square, lower) through rel[]
__global__ __launch_bounds__(256) void k_mf_extend_add(const double* __restrict__ Uc, int bc, const int32_t* __restrict__ rel, double* __restrict__ P, int64_t ld_p, int sp,
                                                       double* __restrict__ Up, int64_t ld_u)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nc = 3 * (int64_t)bc;
    if (t >= nc * nc) return;
    const int64_t j = t / nc, i = t - j * nc;
    if (i < j) return;
    const int64_t gi = 3 * (int64_t)rel[i / 3] + i % 3, gj = 3 * (int64_t)rel[j / 3] + j % 3;  // (rel is increasing: gi >= gj)
    const double v = Uc[i + j * nc];
    if (gj < 3 * (int64_t)sp) P[gi + gj * ld_p] += v;
    else Up[(gi - 3 * sp) + (gj - 3 * sp) * ld_u] += v;
}
// t[3 i + c] = y[3 rows[i] + c] (gather) or y[3 rows[i] + c] += t[3 i + c] (scatter-add)
__global__ __launch_bounds__(256) void k_mf_rows(const int32_t* __restrict__ rows, int nb, double* __restrict__ y, double* __restrict__ t, int scatter)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= 3 * nb) return;
    const int64_t g = 3 * (int64_t)rows[k / 3] + k % 3;
    if (scatter) y[g] += t[k];
    else t[k] = y[g];
}

// adjacency (CSR, no self loops) of the block rows from the pattern's (row, column) pairs
void mf_graph(int64_t nbr, const std::vector<uint32_t>& rows, const std::vector<uint32_t>& cols, std::vector<int64_t>& start, std::vector<int32_t>& adj)
{
    start.assign((size_t)nbr + 1, 0);
    for (size_t k = 0; k < rows.size(); k++)
        if (rows[k] != cols[k]) start[(size_t)rows[k] + 1]++;
    for (int64_t r = 0; r < nbr; r++) start[(size_t)r + 1] += start[(size_t)r];
    adj.assign((size_t)start[(size_t)nbr], 0);
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (size_t k = 0; k < rows.size(); k++)
        if (rows[k] != cols[k]) adj[(size_t)fill[rows[k]]++] = (int32_t)cols[k];
}
struct MfBuilder
{
    const std::vector<int64_t>& start;
    const std::vector<int32_t>& adj;
    std::vector<int32_t> mark;   // id of the vertex set a vertex currently belongs to (-1: removed)
    std::vector<int32_t> level;
    struct Node
    {
        std::vector<int32_t> verts;
        std::vector<int> children;
    };
    std::vector<Node> nodes;
    int next_set = 1;
    const double* xyz = nullptr;  // a position per vertex (NaN: none), or null
    MfBuilder(const std::vector<int64_t>& s, const std::vector<int32_t>& a, int64_t n) : start(s), adj(a), mark((size_t)n, 0), level((size_t)n, -1) {}
    // breadth-first levels of the vertices of set `id` reachable from root; returns them in visiting order
    void bfs(int32_t root, int id, std::vector<int32_t>& out)
    {
        out.clear();
        out.push_back(root);
        level[(size_t)root] = 0;
        for (size_t h = 0; h < out.size(); h++) {
            const int32_t u = out[h];
            for (int64_t j = start[(size_t)u]; j < start[(size_t)u + 1]; j++) {
                const int32_t v = adj[(size_t)j];
                if (mark[(size_t)v] == id && level[(size_t)v] < 0) {
                    level[(size_t)v] = level[(size_t)u] + 1;
                    out.push_back(v);
                }
            }
        }
    }
    // subtrees of the vertex set V (all marked `id`); appends their roots
    void dissect(std::vector<int32_t>& V, int id, std::vector<int>& roots)
    {
        std::vector<int32_t> comp, tmp;
        for (int32_t seed : V) {
            if (mark[(size_t)seed] != id || level[(size_t)seed] >= 0) continue;
            bfs(seed, id, comp);  // one connected component
            if ((int)comp.size() <= MF_LEAF) {
                for (int32_t v : comp) mark[(size_t)v] = -1;
                nodes.push_back(Node{comp, {}});
                roots.push_back((int)nodes.size() - 1);
                continue;
            }
            // with positions: the median plane across the longest extent; the rows of the lower side that touch the upper side are the
            // separator (a layer of mesh nodes: a third of the size of a breadth-first level through a cube, a twentieth of the top fronts' work)
            if (xyz) {
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                bool all = true;
                for (int32_t v : comp) {
                    const double* x = xyz + 3 * (size_t)v;
                    if (!(x[0] == x[0])) {
                        all = false;
                        break;
                    }
                    for (int d = 0; d < 3; d++) {
                        lo[d] = std::min(lo[d], x[d]);
                        hi[d] = std::max(hi[d], x[d]);
                    }
                }
                int ax = 0;
                for (int d = 1; d < 3; d++)
                    if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
                if (all && hi[ax] > lo[ax]) {
                    std::vector<double> key(comp.size());
                    for (size_t k = 0; k < comp.size(); k++) key[k] = xyz[3 * (size_t)comp[k] + ax];
                    std::nth_element(key.begin(), key.begin() + (long)(key.size() / 2), key.end());
                    double med = key[key.size() / 2];
                    if (!(med < hi[ax])) {  // (the upper half is one plane of equal coordinates: cut below it)
                        double below = lo[ax];
                        for (int32_t v : comp) {
                            const double x = xyz[3 * (size_t)v + ax];
                            if (x < med) below = std::max(below, x);
                        }
                        med = below;
                    }
                    const int ida = next_set++, idb = next_set++;
                    std::vector<int32_t> S, A, Bv;
                    for (int32_t v : comp) {
                        mark[(size_t)v] = xyz[3 * (size_t)v + ax] <= med ? ida : idb;
                        level[(size_t)v] = -1;
                    }
                    for (int32_t v : comp) {
                        if (mark[(size_t)v] == idb) {
                            Bv.push_back(v);
                            continue;
                        }
                        bool touches = false;
                        for (int64_t j = start[(size_t)v]; j < start[(size_t)v + 1] && !touches; j++) touches = mark[(size_t)adj[(size_t)j]] == idb;
                        if (touches) S.push_back(v);
                        else A.push_back(v);
                    }
                    if (!Bv.empty() && !S.empty()) {
                        for (int32_t v : S) mark[(size_t)v] = -1;
                        const int me = (int)nodes.size();
                        nodes.push_back(Node{S, {}});
                        roots.push_back(me);
                        std::vector<int> kids;
                        dissect(A, ida, kids);
                        dissect(Bv, idb, kids);
                        nodes[(size_t)me].children = kids;
                        continue;
                    }
                    for (int32_t v : comp) mark[(size_t)v] = id;  // (degenerate: fall through to the level sets)
                }
                for (int32_t v : comp) level[(size_t)v] = 0;  // (as the component search left them)
            }
            // pseudo-peripheral start: twice from the far end
            for (int rep = 0; rep < 2; rep++) {
                const int32_t far = comp.back();
                for (int32_t v : comp) level[(size_t)v] = -1;
                bfs(far, id, tmp);
                comp.swap(tmp);
            }
            const int n_lev = level[(size_t)comp.back()] + 1;
            if (n_lev < 3) {  // no interior level: a clique-like part stays one front
                for (int32_t v : comp) mark[(size_t)v] = -1;
                nodes.push_back(Node{comp, {}});
                roots.push_back((int)nodes.size() - 1);
                continue;
            }
            // the level at which half of the component has been seen (never the first or the last)
            std::vector<int64_t> count((size_t)n_lev, 0);
            for (int32_t v : comp) count[(size_t)level[(size_t)v]]++;
            int m = 1;
            int64_t seen = count[0];
            while (m < n_lev - 2 && seen + count[(size_t)m] < (int64_t)comp.size() / 2) seen += count[(size_t)m++];
            const int ida = next_set++, idb = next_set++;
            std::vector<int32_t> S, A, Bv;
            for (int32_t v : comp) {
                const int l = level[(size_t)v];
                if (l == m) {
                    S.push_back(v);
                    mark[(size_t)v] = -1;
                } else if (l < m) {
                    A.push_back(v);
                    mark[(size_t)v] = ida;
                } else {
                    Bv.push_back(v);
                    mark[(size_t)v] = idb;
                }
                level[(size_t)v] = -1;
            }
            const int me = (int)nodes.size();
            nodes.push_back(Node{S, {}});
            roots.push_back(me);
            std::vector<int> kids;
            dissect(A, ida, kids);
            dissect(Bv, idb, kids);
            nodes[(size_t)me].children = kids;
        }
    }
};
// first-fit arena plan: offset for every request in order, blocks freed when their parent has consumed them
struct ArenaPlan
{
    std::vector<std::pair<size_t, size_t>> free_;  // (offset, size), sorted by offset
    size_t end = 0;
    size_t alloc(size_t n)
    {
        for (size_t k = 0; k < free_.size(); k++)
            if (free_[k].second >= n) {
                const size_t o = free_[k].first;
                if (free_[k].second == n) free_.erase(free_.begin() + (long)k);
                else free_[k] = {o + n, free_[k].second - n};
                return o;
            }
        if (!free_.empty() && free_.back().first + free_.back().second == end) {  // grow the trailing free block
            const size_t o = free_.back().first;
            end = o + n;
            free_.pop_back();
            return o;
        }
        const size_t o = end;
        end += n;
        return o;
    }
    void release(size_t o, size_t n)
    {
        size_t k = 0;
        while (k < free_.size() && free_[k].first < o) k++;
        free_.insert(free_.begin() + (long)k, {o, n});
        if (k + 1 < free_.size() && free_[k].first + free_[k].second == free_[k + 1].first) {
            free_[k].second += free_[k + 1].second;
            free_.erase(free_.begin() + (long)k + 1);
        }
        if (k > 0 && free_[k - 1].first + free_[k - 1].second == free_[k].first) {
            free_[k - 1].second += free_[k].second;
            free_.erase(free_.begin() + (long)k);
        }
    }
};
}  // namespace
void direct_mf_destroy(void* p) { delete static_cast<Multifrontal*>(p); }
namespace {
template <class T>
void mf_upload(Context& c, DevBuf<T>& dst, const std::vector<T>& src)
{
    dst.ensure(std::max<size_t>(src.size(), 1));
    if (!src.empty()) MS_CHECK(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c.stream));
}
// ordering, fronts, index lists, memory plan; returns the gigabytes the factor and the arena need
double mf_analyse(Context& c, Multifrontal& M, const std::vector<uint32_t>& rows, const std::vector<uint32_t>& cols)
{
    const int64_t nbr = c.nbr;
    std::vector<int64_t> start;
    std::vector<int32_t> adj;
    mf_graph(nbr, rows, cols, start, adj);
    MfBuilder Bd(start, adj, nbr);
    // positions of the rows, if the caller handed them over (mistark_dist_set_row_coords; the matrix may live in solver numbering)
    std::vector<double> xyz;
    if ((int64_t)c.sh.coords.size() == 3 * nbr && !c.llt_no_coords) {
        xyz.resize((size_t)(3 * nbr));
        for (int64_t r = 0; r < nbr; r++) {
            const int64_t o = c.perm_active ? (int64_t)c.iperm_h[(size_t)r] : r;
            for (int d = 0; d < 3; d++) xyz[(size_t)(3 * r + d)] = c.sh.coords[(size_t)(3 * o + d)];
        }
        Bd.xyz = xyz.data();
    }
    // hubs first: rows coupled to far more rows than a mesh node (a rigid body under a contact surface) would put everything within two levels
    const int64_t avg = std::max<int64_t>(1, (int64_t)adj.size() / std::max<int64_t>(nbr, 1));
    std::vector<int32_t> hubs, rest;
    for (int64_t r = 0; r < nbr; r++) {
        if (start[(size_t)r + 1] - start[(size_t)r] > std::max<int64_t>(64, 8 * avg)) {
            hubs.push_back((int32_t)r);
            Bd.mark[(size_t)r] = -1;
        } else rest.push_back((int32_t)r);
    }
    std::vector<int> roots;
    Bd.dissect(rest, 0, roots);
    if (!hubs.empty()) {
        Bd.nodes.push_back(MfBuilder::Node{hubs, roots});
        roots.assign(1, (int)Bd.nodes.size() - 1);
    }
    // post-order numbering
    M.fronts.clear();
    M.perm.assign((size_t)nbr, -1);
    std::vector<int> front_of_node(Bd.nodes.size(), -1);
    std::vector<int32_t> front_of((size_t)nbr, -1);
    int64_t next = 0;
    {
        std::vector<std::pair<int, size_t>> stack;  // (node, next child)
        for (int root : roots) {
            stack.push_back({root, 0});
            while (!stack.empty()) {
                auto& top = stack.back();
                const MfBuilder::Node& N = Bd.nodes[(size_t)top.first];
                if (top.second < N.children.size()) {
                    const int ch = N.children[top.second++];
                    stack.push_back({ch, 0});
                    continue;
                }
                MfFront F;
                F.off = next;
                F.s = (int)N.verts.size();
                for (int32_t v : N.verts) {
                    M.perm[(size_t)v] = (int32_t)next;
                    front_of[(size_t)next] = (int)M.fronts.size();
                    next++;
                }
                for (int ch : N.children) {
                    F.children.push_back(front_of_node[(size_t)ch]);
                    M.fronts[(size_t)front_of_node[(size_t)ch]].parent = (int)M.fronts.size();
                }
                front_of_node[(size_t)top.first] = (int)M.fronts.size();
                M.fronts.push_back(F);
                stack.pop_back();
            }
        }
    }
    if (next != nbr) throw Error("DirectLLT: internal: the dissection lost block rows");
    // boundaries, bottom-up (fronts are in post-order: children before parents)
    std::vector<std::vector<int32_t>> bnd(M.fronts.size());
    std::vector<int32_t> inv((size_t)nbr);
    for (int64_t r = 0; r < nbr; r++) inv[(size_t)M.perm[(size_t)r]] = (int32_t)r;
    for (size_t f = 0; f < M.fronts.size(); f++) {
        MfFront& F = M.fronts[f];
        std::vector<int32_t>& B = bnd[f];
        const int64_t end = F.off + F.s;
        for (int64_t i = F.off; i < end; i++) {
            const int32_t r = inv[(size_t)i];
            for (int64_t j = start[(size_t)r]; j < start[(size_t)r + 1]; j++) {
                const int32_t q = M.perm[(size_t)adj[(size_t)j]];
                if (q >= end) B.push_back(q);
            }
        }
        for (int ch : F.children)
            for (int32_t q : bnd[(size_t)ch])
                if (q >= end) B.push_back(q);
        std::sort(B.begin(), B.end());
        B.erase(std::unique(B.begin(), B.end()), B.end());
        F.b = (int)B.size();
        if (F.parent < 0 && F.b != 0) throw Error("DirectLLT: internal: a root front with a boundary");
    }
    // index lists, relative indices, panel offsets, arena plan
    std::vector<int32_t> front_rows, rel, fr_off, fr_s, fr_b;
    std::vector<int64_t> fr_rows_off, fr_panel_off;
    size_t panel_doubles = 0;
    M.max_nf = 0;
    for (size_t f = 0; f < M.fronts.size(); f++) {
        MfFront& F = M.fronts[f];
        F.rows_off = front_rows.size();
        front_rows.insert(front_rows.end(), bnd[f].begin(), bnd[f].end());
        F.panel_off = panel_doubles;
        panel_doubles += (size_t)9 * (size_t)(F.s + F.b) * (size_t)F.s;
        M.max_nf = std::max(M.max_nf, 3 * (F.s + F.b));
        fr_off.push_back((int32_t)F.off);
        fr_s.push_back(F.s);
        fr_b.push_back(F.b);
        fr_rows_off.push_back((int64_t)F.rows_off);
        fr_panel_off.push_back((int64_t)F.panel_off);
    }
    for (size_t f = 0; f < M.fronts.size(); f++) {  // where each boundary row of f sits in its parent
        MfFront& F = M.fronts[f];
        F.rel_off = rel.size();
        if (F.parent < 0) continue;
        const MfFront& P = M.fronts[(size_t)F.parent];
        const std::vector<int32_t>& PB = bnd[(size_t)F.parent];
        for (int32_t q : bnd[f]) {
            if (q < P.off + P.s) rel.push_back((int32_t)(q - P.off));
            else {
                const auto it = std::lower_bound(PB.begin(), PB.end(), q);
                if (it == PB.end() || *it != q) throw Error("DirectLLT: internal: a child's boundary row is missing in its parent");
                rel.push_back((int32_t)(P.s + (it - PB.begin())));
            }
        }
    }
    ArenaPlan plan;
    for (size_t f = 0; f < M.fronts.size(); f++) {
        MfFront& F = M.fronts[f];
        const size_t need = (size_t)9 * (size_t)F.b * (size_t)F.b;
        if (need) F.u_off = plan.alloc(need);
        for (int ch : F.children) {
            const MfFront& C = M.fronts[(size_t)ch];
            if (C.b) plan.release(C.u_off, (size_t)9 * (size_t)C.b * (size_t)C.b);
        }
    }
    M.panel_doubles = panel_doubles;
    M.arena_doubles = plan.end;
    mf_upload(c, M.perm_dev, M.perm);
    mf_upload(c, M.front_of_dev, front_of);
    mf_upload(c, M.front_rows_dev, front_rows);
    mf_upload(c, M.rel_dev, rel);
    mf_upload(c, M.fr_off_dev, fr_off);
    mf_upload(c, M.fr_s_dev, fr_s);
    mf_upload(c, M.fr_b_dev, fr_b);
    mf_upload(c, M.fr_rows_off_dev, fr_rows_off);
    mf_upload(c, M.fr_panel_off_dev, fr_panel_off);
    MS_CHECK(hipStreamSynchronize(c.stream));  // (the staging vectors are temporaries)
    return (double)(panel_doubles + plan.end) * 8.0 / 1e9;
}
// [F11; F21] (R rows, mi columns, leading dimension ld) -> [L11; L21]: the panel loop of the block-tridiagonal path
void mf_factor_panel(Context& c, double* T, int64_t ld, int R, int mi, int* info)
{
    auto tiles = [](int x) { return (unsigned)((x + TS - 1) / TS); };
    constexpr int PW = 256;
    for (int pb = 0; pb < mi; pb += PW) {
        const int pw = std::min(PW, mi - pb);
        for (int kb = pb; kb < pb + pw; kb += TS) {
            const int nb = std::min(TS, pb + pw - kb);
            double* Ckk = T + kb + (size_t)kb * ld;
            const double* Pk = T + kb + (size_t)pb * ld;
            if (kb > pb) hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(R - kb), 1), dim3(256), 0, c.stream, R - kb, nb, kb - pb, Pk, ld, Pk, ld, Ckk, ld, 0);
            hipLaunchKernelGGL(k_chol_tile, dim3(1), dim3(256), 0, c.stream, nb, Ckk, ld, info);
            if (R - kb - nb > 0) hipLaunchKernelGGL(k_trsm_tile, dim3(tiles(R - kb - nb)), dim3(64), 0, c.stream, R - kb - nb, nb, (const double*)Ckk, ld, Ckk + nb, ld);
        }
        const int e = pb + pw, nc = mi - e;
        if (nc > 0) {
            const double* Pe = T + e + (size_t)pb * ld;
            hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(R - e), tiles(nc)), dim3(256), 0, c.stream, R - e, nc, pw, Pe, ld, Pe, ld, T + e + (size_t)e * ld, ld, 1);
        }
    }
}
bool direct_llt_multifrontal(Context& c, const double* rhs_dev, double* x_dev, double cap_gb)
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
    if (!c.llt_mf) c.llt_mf = new Multifrontal();
    Multifrontal& M = *static_cast<Multifrontal*>(c.llt_mf);
    if (c.llt_mf_pattern_version != c.pattern_version) {
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
        const double gb = mf_analyse(c, M, rows, cols);
        c.llt_mf_pattern_version = c.pattern_version;
        c.llt_mf_gb = gb;
        if (trace) std::fprintf(stderr, "[llt] multifrontal: %lld block rows, %zu fronts, largest front %d, factor %.2f GB, update arena %.2f GB\n", (long long)nbr, M.fronts.size(), M.max_nf,
                                (double)M.panel_doubles * 8e-9, (double)M.arena_doubles * 8e-9);
        lap("ordering + symbolic");
    }
    if (c.llt_mf_gb > cap_gb)
        throw Error("DirectLLT: the factor of this system (" + std::to_string(n) + " unknowns) needs " + std::to_string((int)c.llt_mf_gb) +
                    " GB (limit MISTARK_DIRECT_MAX_GB = " + std::to_string((int)cap_gb) + "); use the block-Jacobi PCG");
    M.panels.ensure(std::max<size_t>(M.panel_doubles, 1));
    M.arena.ensure(std::max<size_t>(M.arena_doubles, 1));
    M.work.ensure((size_t)M.max_nf + 8);
    c.llt_y.ensure((size_t)n);
    c.llt_info.ensure(2);
    MS_CHECK(hipMemsetAsync(M.panels.p, 0, M.panel_doubles * sizeof(double), c.stream));
    MS_CHECK(hipMemsetAsync(c.llt_info.p, 0, 2 * sizeof(int), c.stream));
    int* status = c.llt_info.p;
    for (int part = 0; part < 2; part++) {
        const BsrPart& mp = c.part[part];
        if (mp.nnzb == 0) continue;
        hipLaunchKernelGGL(k_mf_add, dim3((unsigned)((mp.nnzb * 9 + 255) / 256)), dim3(256), 0, c.stream, mp.vals.p, mp.colw.p, mp.slot_row.p,
                           (part == 0 && mp.n_chunks_static > 0) ? (const uint32_t*)mp.store_slot.p : (const uint32_t*)nullptr, mp.nnzb, (const int32_t*)M.perm_dev.p,
                           (const int32_t*)M.front_of_dev.p, (const int32_t*)M.fr_off_dev.p, (const int32_t*)M.fr_s_dev.p, (const int32_t*)M.fr_b_dev.p,
                           (const int64_t*)M.fr_rows_off_dev.p, (const int64_t*)M.fr_panel_off_dev.p, (const int32_t*)M.front_rows_dev.p, M.panels.p, status);
    }
    lap("fill fronts");
    auto tiles = [](int x) { return (unsigned)((x + TS - 1) / TS); };
    for (size_t f = 0; f < M.fronts.size(); f++) {
        const MfFront& F = M.fronts[f];
        const int nf = 3 * (F.s + F.b), ms = 3 * F.s, mbd = 3 * F.b;
        double* P = M.panels.p + F.panel_off;
        double* U = M.arena.p + F.u_off;
        if (mbd > 0) MS_CHECK(hipMemsetAsync(U, 0, (size_t)mbd * mbd * sizeof(double), c.stream));
        for (int ch : F.children) {
            const MfFront& C = M.fronts[(size_t)ch];
            if (C.b == 0) continue;
            const int64_t nc = 3 * (int64_t)C.b;
            hipLaunchKernelGGL(k_mf_extend_add, dim3((unsigned)((nc * nc + 255) / 256)), dim3(256), 0, c.stream, (const double*)(M.arena.p + C.u_off), C.b,
                               (const int32_t*)(M.rel_dev.p + C.rel_off), P, (int64_t)nf, F.s, U, (int64_t)mbd);
        }
        mf_factor_panel(c, P, nf, nf, ms, status + 1);
        if (mbd > 0)
            hipLaunchKernelGGL(k_gemm_nt_sub, dim3(tiles(mbd), tiles(mbd)), dim3(256), 0, c.stream, mbd, mbd, ms, (const double*)(P + ms), (int64_t)nf, (const double*)(P + ms), (int64_t)nf, U,
                               (int64_t)mbd, 1);
    }
    lap("factorisation");
    int h[2] = {1, 1};
    fetch(c, h, status, 2 * sizeof(int));
    if (h[0] == 2) throw Error("DirectLLT: internal: a matrix entry fell outside its front");
    if (h[1] != 0) return false;  // a non-positive pivot: SimplicialLLT's info() != Success
    // ---- L y = P b (up the tree), L^T z = y (down), x = P^T z
    double* y = c.llt_y.p;
    double* w = M.work.p;
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((3 * nbr + 255) / 256)), dim3(256), 0, c.stream, rhs_dev, (const int32_t*)M.perm_dev.p, nbr, true, y);
    for (size_t f = 0; f < M.fronts.size(); f++) {
        const MfFront& F = M.fronts[f];
        const int nf = 3 * (F.s + F.b), ms = 3 * F.s, mbd = 3 * F.b;
        const double* P = M.panels.p + F.panel_off;
        double* ys = y + 3 * F.off;
        if (mbd > 0) MS_CHECK(hipMemsetAsync(w, 0, (size_t)mbd * sizeof(double), c.stream));
        for (int kb = 0; kb < ms; kb += TS) {
            const int nb = std::min(TS, ms - kb);
            const double* Ckk = P + kb + (size_t)kb * nf;
            hipLaunchKernelGGL(k_trsv_tile, dim3(1), dim3(64), 0, c.stream, nb, Ckk, (int64_t)nf, ys + kb, 0);
            const int below = ms - kb - nb;
            if (below > 0) hipLaunchKernelGGL(k_gemv_sub, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, c.stream, below, nb, Ckk + nb, (int64_t)nf, (const double*)(ys + kb), ys + kb + nb);
            if (mbd > 0) hipLaunchKernelGGL(k_gemv_sub, dim3((unsigned)((mbd + 255) / 256)), dim3(256), 0, c.stream, mbd, nb, P + ms + (size_t)kb * nf, (int64_t)nf, (const double*)(ys + kb), w);
        }
        if (mbd > 0) hipLaunchKernelGGL(k_mf_rows, dim3((unsigned)((mbd + 255) / 256)), dim3(256), 0, c.stream, (const int32_t*)(M.front_rows_dev.p + F.rows_off), F.b, y, w, 1);
    }
    for (size_t fi = M.fronts.size(); fi-- > 0;) {
        const MfFront& F = M.fronts[fi];
        const int nf = 3 * (F.s + F.b), ms = 3 * F.s, mbd = 3 * F.b;
        const double* P = M.panels.p + F.panel_off;
        double* ys = y + 3 * F.off;
        if (mbd > 0) hipLaunchKernelGGL(k_mf_rows, dim3((unsigned)((mbd + 255) / 256)), dim3(256), 0, c.stream, (const int32_t*)(M.front_rows_dev.p + F.rows_off), F.b, y, w, 0);
        for (int kb = (ms - 1) / TS * TS; kb >= 0; kb -= TS) {
            const int nb = std::min(TS, ms - kb);
            const double* Ckk = P + kb + (size_t)kb * nf;
            const int below = ms - kb - nb;
            if (below > 0) hipLaunchKernelGGL(k_gemv_t_sub, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c.stream, below, nb, Ckk + nb, (int64_t)nf, (const double*)(ys + kb + nb), ys + kb);
            if (mbd > 0) hipLaunchKernelGGL(k_gemv_t_sub, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c.stream, mbd, nb, P + ms + (size_t)kb * nf, (int64_t)nf, (const double*)w, ys + kb);
            hipLaunchKernelGGL(k_trsv_tile, dim3(1), dim3(64), 0, c.stream, nb, Ckk, (int64_t)nf, ys + kb, 1);
        }
    }
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((3 * nbr + 255) / 256)), dim3(256), 0, c.stream, (const double*)y, (const int32_t*)M.perm_dev.p, nbr, false, x_dev);
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
