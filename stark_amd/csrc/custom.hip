// custom.hip — potentials without a hand-written kernel: the energy arrives as SymX's own straight-line op sequence
// (symx::Sequence, symx/src/compile/Sequence.h:24-41; op semantics as emitted by symx/src/compile/Compilation.cpp:381-469) and is
// INTERPRETED on hyper-dual numbers, one lane per (element, i <= j) pair of local DoFs — the same decomposition as the generic
// kernels of kernels.hip, with the expression read from memory instead of compiled in. This is the fallback that keeps user-defined
// potentials (`README.md:109-126`, examples/main.cpp magnetic_deformables_implicit) on the GPU path; it is not
// fast (register file in scratch memory), and every potential a stock stark::Simulation registers has a compiled kernel instead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "engine.hpp"
#include "hdual.hpp"

namespace mistark {

// symx::ExprType (symx/src/symbol/Expr.h:12-43)
enum : int32_t { OP_ZERO = 0, OP_ONE = 1, OP_BRANCH = 2, OP_CONST = 4, OP_SYMBOL = 5, OP_ADD = 6, OP_SUB = 7, OP_MUL = 8, OP_RECIP = 9, OP_POWN = 10, OP_POWF = 11, OP_SQRT = 12,
                 OP_LN = 13, OP_LOG10 = 14, OP_EXP = 15, OP_SIN = 16, OP_COS = 17, OP_TAN = 18, OP_ASIN = 19, OP_ACOS = 20, OP_ATAN = 21, OP_PRINT = 22 };
constexpr int CUSTOM_MAX_REGS = 256;  // live temporaries after register allocation (EnergyTriangleStrain needs 70, the rigid-rigid contact potentials 206)
constexpr int CUSTOM_MAX_IN = 96;     // gathered inputs per element

struct CustomProgram
{
    // program after register allocation: rows {type, dst, a, b, cond}; value index < n_in = input, otherwise register n_in + r
    std::vector<int32_t> ops, cops;
    std::vector<double> consts, cconsts;
    int n_in = 0, n_regs = 0, n_cregs = 0;
    std::vector<int32_t> strides;
    DevBuf<int32_t> d_ops, d_cops, d_strides, d_in_dof;
    DevBuf<double> d_consts, d_cconsts, d_sum;
    // summation loop (MappedWorkspace::add_for_each, symx/src/compile/MappedWorkspace.h:123-130,519-553): inputs [sum_first, sum_first + sum_stride)
    // take the rows of sum_data one after the other and the element's energy / gradient / Hessian is the SUM over the rows
    // (CompiledInLoop_run.h:375-400), projected once like any other element
    int sum_first = -1, sum_stride = 0, sum_n = 0;
    std::vector<double> sum_data;
    bool uploaded = false;
};

namespace {
struct ProgDev
{
    const int32_t* ops;
    const double* consts;
    int n_ops;
    const int32_t* cops;
    const double* cconsts;
    int n_cops;
    const int32_t* strides;
    const int32_t* in_dof;  // per gathered input: local DoF component (3 * block + c) or -1
    int n_bind, n_in, NB;
    int sum_first, sum_stride, sum_n;  // summation loop: sum_n == 0 = none
    const double* sum_data;
};

__device__ __forceinline__ HDual powi(const HDual& x, int n)
{
    if (n == 0) return HDual(1.0);
    const double p2 = ::pow(x.v, (double)(n - 2)), p1 = p2 * x.v;  // x^(n-2), x^(n-1)
    if (n == 1) return x;
    if (n == 2) return x * x;
    return chain(x, p1 * x.v, n * p1, (double)n * (n - 1) * p2);
}
__device__ __forceinline__ HDual powf_h(const HDual& x, const HDual& y)
{
    // x^y = exp(y ln x)
    const HDual t = y * log(x);
    const double e = ::exp(t.v);
    return chain(t, e, e, e);
}
// Runs one op sequence; returns the value bound to output 0.
__device__ HDual run_program(const int32_t* __restrict__ ops, const double* __restrict__ consts, int n_ops, const double* in, const int32_t* __restrict__ in_dof, int n_in, int si, int sj)
{
    HDual reg[CUSTOM_MAX_REGS];
    auto get = [&](int idx) -> HDual {
        if (idx < n_in) {
            const int d = in_dof[idx];
            return HDual(in[idx], (d >= 0 && d == si) ? 1.0 : 0.0, (d >= 0 && d == sj) ? 1.0 : 0.0, 0.0);
        }
        return reg[idx - n_in];
    };
    HDual out(0.0);
    // if / else / endif markers: a stack of (parent active, branch taken) bits; ops of an inactive region are skipped per lane
    uint32_t parent = 0, taken = 0;
    int depth = 0;
    bool active = true;
    for (int k = 0; k < n_ops; k++) {
        const int32_t* op = ops + 5 * k;
        const int type = op[0], dst = op[1], a = op[2], b = op[3], cond = op[4];
        if (type == OP_BRANCH) {
            if (cond == -2) {  // endif
                depth--;
                active = (parent >> depth) & 1u;
            } else if (a == 0) {  // if (cond > 0)
                const bool t = active && get(cond).v > 0.0;
                parent = (parent & ~(1u << depth)) | ((active ? 1u : 0u) << depth);
                taken = (taken & ~(1u << depth)) | ((t ? 1u : 0u) << depth);
                depth++;
                active = t;
            } else {  // else
                active = ((parent >> (depth - 1)) & 1u) && !((taken >> (depth - 1)) & 1u);
            }
            continue;
        }
        if (!active) continue;
        HDual r;
        switch (type) {
            case OP_SYMBOL: out = get(a); continue;  // out[dst] = value (a single output: the energy)
            case OP_ZERO: r = HDual(0.0); break;
            case OP_ONE: r = HDual(1.0); break;
            case OP_CONST: r = HDual(consts[k]); break;
            case OP_ADD: r = get(a) + get(b); break;
            case OP_SUB: r = get(a) - get(b); break;
            case OP_MUL: r = get(a) * get(b); break;
            case OP_RECIP: r = inv(get(a)); break;
            case OP_POWN: r = powi(get(a), b); break;
            case OP_POWF: r = powf_h(get(a), get(b)); break;
            case OP_SQRT: r = sqrt(get(a)); break;
            case OP_LN: {
                const HDual x = get(a);
                r = x.v <= 0.0 ? HDual(-INFINITY) : log(x);
                break;
            }
            case OP_LOG10: {
                const HDual x = get(a);
                r = x.v <= 0.0 ? HDual(-INFINITY) : (1.0 / ::log(10.0)) * log(x);
                break;
            }
            case OP_EXP: {
                const HDual x = get(a);
                const double e = ::exp(x.v);
                r = chain(x, e, e, e);
                break;
            }
            case OP_SIN: r = sin(get(a)); break;
            case OP_COS: r = cos(get(a)); break;
            case OP_TAN: {
                const HDual x = get(a);
                const double t = ::tan(x.v), s = 1.0 + t * t;
                r = chain(x, t, s, 2.0 * t * s);
                break;
            }
            case OP_ASIN: {
                const HDual x = get(a);
                const double s = 1.0 / ::sqrt(1.0 - x.v * x.v);
                r = chain(x, ::asin(x.v), s, x.v * s * s * s);
                break;
            }
            case OP_ACOS: r = acos(get(a)); break;
            case OP_ATAN: r = atan(get(a)); break;
            default: r = HDual(0.0); break;  // Print
        }
        reg[dst - n_in] = r;
    }
    return out;
}
__device__ __forceinline__ void gather_custom(const PotArgs& a, const ProgDev& p, int e, double* in)
{
    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;
    int o = 0;
    for (int b = 0; b < p.n_bind; b++) {
        const int S = p.strides[b], col = a.conn_col[b];
        const double* src = a.arr[b] + (col < 0 ? 0 : (size_t)ce[col]) * S;
        for (int c = 0; c < S; c++) in[o + c] = src[c];
        o += S;
    }
}
// mode 0: energy only (one lane per element); mode 1: energy + gradient (one lane per (element, i)); mode 2: + Hessian ((element, i <= j))
template <int MODE>
__global__ __launch_bounds__(256) void k_eval_custom(PotArgs a, ProgDev p, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)
{
    const int NB = p.NB, n = 3 * NB, NP = MODE == 0 ? 1 : (MODE == 1 ? n : n * (n + 1) / 2);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)a.e_count * NP) return;
    const int le = (int)(t / NP);
    const int e = a.elem_list ? (int)a.elem_list[le] : a.e_begin + le;
    const int pe = a.elem_list ? le : e;  // position in the pools (kernels.hip: pool_of)
    int rem = (int)(t - (long long)le * NP);
    const bool first = rem == 0;
    int i = -1, j = -1;
    if (MODE == 1) i = j = rem;
    if (MODE == 2) {
        i = 0;
        while (rem >= n - i) {
            rem -= n - i;
            i++;
        }
        j = i + rem;
    }
    double in[CUSTOM_MAX_IN];
    gather_custom(a, p, e, in);
    bool on = true;
    if (p.n_cops > 0) {  // SecondOrderCompiledPotential.cpp:185-197: element active iff the condition's value is > 0
        if (p.sum_n == 0) {
            on = run_program(p.cops, p.cconsts, p.n_cops, in, p.in_dof, p.n_in, -1, -1).v > 0.0;
        } else {
            // the condition is compiled over the same workspace, so the reference runs IT through the summation loop as well
            // (CompiledInLoop_run.h:375-400): its value is the sum over the rows of the summation data, accumulated in row order
            double cv = 0.0;
            for (int it = 0; it < p.sum_n; it++) {
                for (int c = 0; c < p.sum_stride; c++) in[p.sum_first + c] = p.sum_data[(size_t)it * p.sum_stride + c];
                cv += run_program(p.cops, p.cconsts, p.n_cops, in, p.in_dof, p.n_in, -1, -1).v;
            }
            on = cv > 0.0;
        }
    }
    HDual r(0.0);
    if (on) {
        if (p.sum_n == 0) {
            r = run_program(p.ops, p.consts, p.n_ops, in, p.in_dof, p.n_in, i, j);
        } else {
            for (int it = 0; it < p.sum_n; it++) {  // the reference accumulates the iterations in this order (CompiledInLoop_run.h:376)
                for (int c = 0; c < p.sum_stride; c++) in[p.sum_first + c] = p.sum_data[(size_t)it * p.sum_stride + c];
                r = r + run_program(p.ops, p.consts, p.n_ops, in, p.in_dof, p.n_in, i, j);
            }
        }
    }
    if (MODE == 2) {
        const int ba = i / 3, ii = i - 3 * ba, bb = j / 3, jj = j - 3 * bb;
        elemH[((size_t)(ba * NB + bb) * a.n_pool + pe) * 9 + ii * 3 + jj] = r.ab;
        elemH[((size_t)(bb * NB + ba) * a.n_pool + pe) * 9 + jj * 3 + ii] = r.ab;
    }
    if (MODE >= 1 && i == j && on) {
        const int ba = i / 3, ii = i - 3 * ba;
        const int node = a.conn[(size_t)e * a.conn_stride + a.dof_col[ba]];
        if (a.hot_base[ba] >= 0) atomicAdd(&a.grad_hot[((size_t)(blockIdx.x & (HOT_WAYS - 1)) * a.n_hot + a.hot_base[ba] + node) * 3 + ii], r.a);  // (k_eval_pgh)
        else atomicAdd(&grad[3 * (size_t)(a.dof_row_off[ba] + node) + ii], r.a);
    }
    if (first) elemE[pe] = r.v;
}

// Linear-scan register allocation of the temporaries of one op sequence (values >= n_in). A value defined in both arms of a branch
// gets one register: its live range runs from its first definition to its last use.
void allocate_registers(std::vector<int32_t>& ops, int n_ops, int n_in, int& n_regs, const std::string& name)
{
    int max_val = n_in;
    auto reads = [&](const int32_t* op, int* out) {  // value indices an op reads
        int k = 0;
        const int type = op[0];
        if (type == OP_BRANCH) {
            if (op[4] != -2 && op[2] == 0) out[k++] = op[4];
        } else if (type == OP_SYMBOL || type == OP_RECIP || type == OP_POWN || (type >= OP_SQRT && type <= OP_PRINT)) {
            out[k++] = op[2];
        } else if (type == OP_ADD || type == OP_SUB || type == OP_MUL || type == OP_POWF) {
            out[k++] = op[2];
            out[k++] = op[3];
        }
        return k;
    };
    auto writes = [&](const int32_t* op) { return (op[0] == OP_BRANCH || op[0] == OP_SYMBOL) ? -1 : op[1]; };
    for (int k = 0; k < n_ops; k++) {
        const int32_t* op = &ops[5 * k];
        int r[2];
        const int nr = reads(op, r);
        for (int q = 0; q < nr; q++) {
            if (r[q] < 0) throw Error("custom potential '" + name + "': op " + std::to_string(k) + " reads an invalid value");
            max_val = std::max(max_val, r[q] + 1);
        }
        const int w = writes(op);
        if (w >= 0) {
            if (w < n_in) throw Error("custom potential '" + name + "': op " + std::to_string(k) + " overwrites an input");
            max_val = std::max(max_val, w + 1);
        }
        if (op[0] == OP_SYMBOL && op[1] != 0) throw Error("custom potential '" + name + "': exactly one output (the energy) is expected");
        if (op[0] == 3 || op[0] < 0 || op[0] > OP_PRINT) throw Error("custom potential '" + name + "': unknown op type " + std::to_string(op[0]));
    }
    const int nt = max_val - n_in;
    std::vector<int> first(nt, -1), last(nt, -1);
    for (int k = 0; k < n_ops; k++) {
        const int32_t* op = &ops[5 * k];
        int r[2];
        const int nr = reads(op, r);
        for (int q = 0; q < nr; q++)
            if (r[q] >= n_in) {
                if (first[r[q] - n_in] < 0) throw Error("custom potential '" + name + "': value used before its definition");
                last[r[q] - n_in] = k;
            }
        const int w = writes(op);
        if (w >= n_in) {
            if (first[w - n_in] < 0) first[w - n_in] = k;
            last[w - n_in] = std::max(last[w - n_in], k);
        }
    }
    // scan in program order: free the registers whose value died, take the lowest free one for a new value
    std::vector<int> reg_of(nt, -1);
    std::vector<int> free_regs;
    std::vector<std::vector<int>> dying(n_ops + 1);
    for (int t = 0; t < nt; t++)
        if (first[t] >= 0) dying[last[t]].push_back(t);
    n_regs = 0;
    std::vector<int32_t> out = ops;
    for (int k = 0; k < n_ops; k++) {
        int32_t* op = &out[5 * k];
        int r[2];
        const int nr = reads(&ops[5 * k], r);
        (void)nr;
        const int w = writes(&ops[5 * k]);
        if (w >= n_in && reg_of[w - n_in] < 0) {
            int reg;
            if (!free_regs.empty()) {
                std::sort(free_regs.begin(), free_regs.end(), std::greater<int>());
                reg = free_regs.back();
                free_regs.pop_back();
            } else reg = n_regs++;
            reg_of[w - n_in] = reg;
        }
        auto map = [&](int32_t v) { return v >= n_in ? (int32_t)(n_in + reg_of[v - n_in]) : v; };
        const int type = op[0];
        if (type == OP_BRANCH) {
            if (op[4] != -2 && op[2] == 0) op[4] = map(op[4]);
        } else {
            if (type == OP_SYMBOL || type == OP_RECIP || type == OP_POWN || (type >= OP_SQRT && type <= OP_PRINT)) op[2] = map(op[2]);
            if (type == OP_ADD || type == OP_SUB || type == OP_MUL || type == OP_POWF) {
                op[2] = map(op[2]);
                op[3] = map(op[3]);
            }
            if (w >= 0) op[1] = map(op[1]);
        }
        // a value read for the last time by this op may give its register to the NEXT definition (not to this op's own result:
        // the interpreter reads both operands before it writes)
        for (int t : dying[k]) free_regs.push_back(reg_of[t]);
    }
    if (n_regs > CUSTOM_MAX_REGS)
        throw Error("custom potential '" + name + "': the expression needs " + std::to_string(n_regs) + " live temporaries, the interpreter has " + std::to_string(CUSTOM_MAX_REGS));
    ops.swap(out);
}
}  // namespace

std::shared_ptr<CustomProgram> make_custom_program(const std::string& name, const int32_t* strides, int n_bindings, const int32_t* ops, const double* consts, int n_ops, int n_inputs,
                                                   const int32_t* cond_ops, const double* cond_consts, int n_cond_ops)
{
    auto P = std::make_shared<CustomProgram>();
    int sum = 0;
    for (int b = 0; b < n_bindings; b++) sum += strides[b];
    if (sum != n_inputs) throw Error("custom potential '" + name + "': the bindings provide " + std::to_string(sum) + " inputs, the op sequence expects " + std::to_string(n_inputs));
    if (n_inputs > CUSTOM_MAX_IN) throw Error("custom potential '" + name + "': more than " + std::to_string(CUSTOM_MAX_IN) + " inputs per element");
    if (n_ops <= 0 || !ops || !consts) throw Error("custom potential '" + name + "': empty op sequence");
    P->n_in = n_inputs;
    P->strides.assign(strides, strides + n_bindings);
    P->ops.assign(ops, ops + 5 * (size_t)n_ops);
    P->consts.assign(consts, consts + n_ops);
    allocate_registers(P->ops, n_ops, n_inputs, P->n_regs, name);
    if (n_cond_ops > 0) {
        P->cops.assign(cond_ops, cond_ops + 5 * (size_t)n_cond_ops);
        P->cconsts.assign(cond_consts, cond_consts + n_cond_ops);
        allocate_registers(P->cops, n_cond_ops, n_inputs, P->n_cregs, name + " (condition)");
    }
    return P;
}

void custom_program_set_summation(CustomProgram& G, const std::string& name, int first_input, int stride, int n_iterations, const double* data)
{
    if (stride <= 0 || n_iterations <= 0 || !data) throw Error("custom potential '" + name + "': summation needs stride > 0, iterations > 0 and data");
    if (first_input < 0 || first_input + stride > G.n_in) throw Error("custom potential '" + name + "': summation inputs outside the potential's " + std::to_string(G.n_in) + " inputs");
    G.sum_first = first_input;
    G.sum_stride = stride;
    G.sum_n = n_iterations;
    G.sum_data.assign(data, data + (size_t)stride * n_iterations);
    G.uploaded = false;
}

// Evaluation of a custom potential (called by kernels.hip: launch_eval_kind for P.kind == KIND_CUSTOM)
void launch_eval_custom(Context& c, Potential& P, int mode)
{
    if (P.args.e_count == 0) return;
    CustomProgram& G = *P.prog;
    if (!G.uploaded) {
        auto up_i = [&](DevBuf<int32_t>& d, const std::vector<int32_t>& h) {
            d.ensure(std::max<size_t>(h.size(), 1));
            if (!h.empty()) MS_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
        };
        auto up_d = [&](DevBuf<double>& d, const std::vector<double>& h) {
            d.ensure(std::max<size_t>(h.size(), 1));
            if (!h.empty()) MS_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c.stream));
        };
        up_i(G.d_ops, G.ops);
        up_i(G.d_cops, G.cops);
        up_i(G.d_strides, G.strides);
        up_d(G.d_consts, G.consts);
        up_d(G.d_cconsts, G.cconsts);
        up_d(G.d_sum, G.sum_data);
        MS_CHECK(hipStreamSynchronize(c.stream));
        G.uploaded = true;
    }
    // input -> local DoF component, in the block order prepare() gave the potential (DoF sets in registration order, then binding order)
    std::vector<int32_t> in_dof((size_t)G.n_in, -1);
    {
        int blk = 0;
        for (int set = 0; set < (int)c.dof_sets.size(); set++) {
            int o = 0;
            for (size_t b = 0; b < P.bindings.size(); b++) {
                if (c.arrays[P.bindings[b].array].dof_set == set) {
                    for (int k = 0; k < 3; k++) in_dof[(size_t)o + k] = 3 * blk + k;
                    blk++;
                }
                o += G.strides[b];
            }
        }
    }
    // (the summation loop overwrites its inputs with constants in every iteration: they cannot be DoFs, whose derivative seeds in_dof carries)
    for (int k = 0; G.sum_n > 0 && k < G.sum_stride; k++)
        if (in_dof[(size_t)(G.sum_first + k)] >= 0)
            throw Error("custom potential '" + P.name + "': the summation inputs [" + std::to_string(G.sum_first) + ", " + std::to_string(G.sum_first + G.sum_stride) +
                        ") overlap a binding on a DoF set (input " + std::to_string(G.sum_first + k) + ")");
    G.d_in_dof.ensure(std::max<size_t>(in_dof.size(), 1));
    MS_CHECK(hipMemcpyAsync(G.d_in_dof.p, in_dof.data(), in_dof.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));  // (in_dof is a temporary)
    const ProgDev pd{G.d_ops.p, G.d_consts.p, (int)G.consts.size(), G.d_cops.p, G.d_cconsts.p, (int)G.cconsts.size(), G.d_strides.p, G.d_in_dof.p, (int)G.strides.size(), G.n_in, P.NB,
                     G.sum_first, G.sum_stride, G.sum_n, G.d_sum.p};
    double* E = c.elemE.p + P.e_off;
    const int n = 3 * P.NB;
    auto grid = [&](long long threads) { return dim3((unsigned)((threads + 255) / 256)); };
    if (mode == MISTARK_EVAL_P)
        hipLaunchKernelGGL(k_eval_custom<0>, grid(P.args.e_count), dim3(256), 0, c.stream, P.args, pd, E, (double*)nullptr, (double*)nullptr);
    else if (mode == MISTARK_EVAL_P_G)
        hipLaunchKernelGGL(k_eval_custom<1>, grid((long long)P.args.e_count * n), dim3(256), 0, c.stream, P.args, pd, E, (double*)nullptr, c.grad.p);
    else
        hipLaunchKernelGGL(k_eval_custom<2>, grid((long long)P.args.e_count * (n * (n + 1) / 2)), dim3(256), 0, c.stream, P.args, pd, E, c.elemH.p + P.h_off, c.grad.p);
}

}  // namespace mistark
