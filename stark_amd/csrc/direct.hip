// direct.hip — the reference's DirectLLT alternative to the block-Jacobi PCG (symx/src/solver/NewtonsMethod.cpp:395-418: the float BSR
// Hessian as double triplets, Eigen::SimplicialLLT, du = A^-1 (-grad); any non-positive pivot = "solve failed").
//
// The reference uses it for small stiff rigid-body problems (tests/rb_constraints.cpp:40). Here it is a dense Cholesky on the device for
// systems of up to MAX_DIRECT_DOFS unknowns: one workgroup, right-looking, the trailing update spread over its threads. Large FEM systems
// belong to the PCG; asking for DirectLLT beyond the limit is an error, not a silent switch.
#include <hip/hip_runtime.h>

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
}  // namespace

// du = A^-1 rhs with A = A_static + A_dynamic as assembled. Returns false when the factorisation meets a non-positive pivot.
bool direct_llt(Context& c, const double* rhs_dev, double* x_dev)
{
    if (!c.have_matrix) throw Error("direct_llt: matrix not assembled");
    if (c.world > 1) throw Error("DirectLLT is a single-rank solver (a sharded context holds its own rows only); use the block-Jacobi PCG");
    const int n = (int)c.ndofs;
    if (c.ndofs > MAX_DIRECT_DOFS)
        throw Error("DirectLLT is a dense factorisation for small systems (<= " + std::to_string(MAX_DIRECT_DOFS) + " unknowns, this one has " + std::to_string(c.ndofs) +
                    "); use the block-Jacobi PCG");
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
