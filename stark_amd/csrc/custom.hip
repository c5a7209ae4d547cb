// custom.hip — potentials without a hand-written kernel: the energy arrives as SymX's own straight-line op sequence
// (symx::Sequence, symx/src/compile/Sequence.h:24-41; op semantics as emitted by symx/src/compile/Compilation.cpp:381-469) and is
// INTERPRETED on hyper-dual numbers, one lane per (element, i <= j) pair of local DoFs — the same decomposition as the generic
// kernels of kernels.hip, with the expression read from memory instead of compiled in. This is the fallback that keeps user-defined
// potentials (`README.md:109-126`, examples/main.cpp magnetic_deformables_implicit) on the GPU path; it is not
// fast (register file in scratch memory), and every potential a stock stark::Simulation registers has a compiled kernel instead.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <chrono>
#include <cstring>

#include "engine.hpp"
#include "hdual.hpp"
#include "custom_math.hpp"

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
    // kernels emitted for this program (hipRTC; see "the emitter" below): built at the first evaluation, rebuilt when what the source depends on
    // (DoF bindings, block count, summation layout) changes; `rtc_failed` keeps the interpreter for good after a failed build
    std::shared_ptr<struct RtcKernels> rtc;
    std::string rtc_key;
    bool rtc_failed = false;
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
            case OP_POWN: r = cop_pown(get(a), b); break;
            case OP_POWF: r = cop_powf(get(a), get(b)); break;
            case OP_SQRT: r = sqrt(get(a)); break;
            case OP_LN: r = cop_ln(get(a)); break;
            case OP_LOG10: r = cop_log10(get(a)); break;
            case OP_EXP: r = cop_exp(get(a)); break;
            case OP_SIN: r = sin(get(a)); break;
            case OP_COS: r = cos(get(a)); break;
            case OP_TAN: r = cop_tan(get(a)); break;
            case OP_ASIN: r = cop_asin(get(a)); break;
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
    if (first) elemE[pe] = energy_here(a, e) ? r.v : 0.0;  // (sharded runs: an interface element is evaluated by every rank that owns one of its rows; its energy counts once)
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


// ======================================================================================================================================
// The emitter ("a small expression -> HIP emitter consuming SymX's op Sequence ... compiled with hipcc / hipRTC"; the scalar
// emitter it mirrors: symx/src/compile/Compilation.cpp:381-469). The op sequence of a user-defined potential becomes straight-line HIP source
// — one statement per op, the temporaries local variables, if / else / endif as real branches, the gather of the bound arrays unrolled with
// the strides and DoF seeds known at emission — compiled for gfx950 by hipRTC at the potential's first evaluation and cached on disk by a hash
// of the source (MISTARK_RTC_CACHE, default /tmp/mistark_rtc_cache_<uid>; the reference caches its JIT-compiled .so files the same way,
// Compilation.cpp:243-336). Same decomposition (one lane per (element, i <= j)), same hyper-dual operations (custom_math.hpp is compiled into
// both) and same output layout as the interpreter k_eval_custom, which stays as the fallback (no hipRTC library, a failed build, programs
// beyond MISTARK_RTC_MAX_OPS, option custom_rtc = 0) and as the cross-check (tests/test_gpu_custom_rtc.py).
// ======================================================================================================================================
struct RtcKernels
{
    hipModule_t mod = nullptr;
    hipFunction_t fn[3] = {nullptr, nullptr, nullptr};
    ~RtcKernels()
    {
        if (mod) (void)hipModuleUnload(mod);
    }
};
namespace {
static const char* RTC_POTARGS_SRC =
#include "rtc_potargs.inc"
    ;
static const char* RTC_HDUAL_SRC =
#include "rtc_hdual.inc"
    ;
static const char* RTC_MATH_SRC =
#include "rtc_custom_math.inc"
    ;
struct HipRtc
{
    void* lib = nullptr;
    bool tried = false;
    int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*CompileProgram)(void*, int, const char**) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
    int (*Version)(int*, int*) = nullptr;  // (optional: part of the cache key)
};
HipRtc& hiprtc()
{
    static HipRtc r;
    if (!r.tried) {
        r.tried = true;
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            bool ok = true;
            auto sym = [&](const char* n) {
                void* p = dlsym(r.lib, n);
                if (!p) ok = false;
                return p;
            };
            r.CreateProgram = (int (*)(void**, const char*, const char*, int, const char**, const char**))sym("hiprtcCreateProgram");
            r.CompileProgram = (int (*)(void*, int, const char**))sym("hiprtcCompileProgram");
            r.GetProgramLogSize = (int (*)(void*, size_t*))sym("hiprtcGetProgramLogSize");
            r.GetProgramLog = (int (*)(void*, char*))sym("hiprtcGetProgramLog");
            r.GetCodeSize = (int (*)(void*, size_t*))sym("hiprtcGetCodeSize");
            r.GetCode = (int (*)(void*, char*))sym("hiprtcGetCode");
            r.DestroyProgram = (int (*)(void**))sym("hiprtcDestroyProgram");
            if (!ok) {
                dlclose(r.lib);
                r.lib = nullptr;
            } else r.Version = (int (*)(int*, int*))dlsym(r.lib, "hiprtcVersion");
        }
    }
    return r;
}
std::string rtc_double(double v)
{
    // exact bits (the interpreter reads the same doubles from memory)
    long long b;
    std::memcpy(&b, &v, 8);
    char buf[64];
    std::snprintf(buf, sizeof(buf), "__longlong_as_double(%lldLL)", b);
    return buf;
}
// one op sequence as a device function: HDual NAME(const double (&in)[n_in], int si, int sj)
void emit_program(std::string& out, const char* fname, const std::vector<int32_t>& ops, const std::vector<double>& consts, int n_in, int n_regs, const std::vector<int32_t>& in_dof)
{
    const int n_ops = (int)consts.size();
    std::vector<char> used((size_t)n_in, 0);
    auto mark = [&](int v) {
        if (v >= 0 && v < n_in) used[(size_t)v] = 1;
    };
    for (int k = 0; k < n_ops; k++) {
        const int32_t* op = &ops[5 * (size_t)k];
        const int type = op[0];
        if (type == OP_BRANCH) {
            if (op[4] != -2 && op[2] == 0) mark(op[4]);
        } else if (type == OP_SYMBOL || type == OP_RECIP || type == OP_POWN || (type >= OP_SQRT && type <= OP_PRINT)) mark(op[2]);
        else if (type == OP_ADD || type == OP_SUB || type == OP_MUL || type == OP_POWF) {
            mark(op[2]);
            mark(op[3]);
        }
    }
    out += "__device__ __forceinline__ HDual ";
    out += fname;
    out += "(const double (&in)[" + std::to_string(std::max(n_in, 1)) + "], const int si, const int sj)\n{\n";
    for (int i = 0; i < n_in; i++) {
        if (!used[(size_t)i]) continue;
        const int d = in_dof[(size_t)i];
        if (d >= 0) out += "    const HDual x" + std::to_string(i) + "(in[" + std::to_string(i) + "], si == " + std::to_string(d) + " ? 1.0 : 0.0, sj == " + std::to_string(d) + " ? 1.0 : 0.0, 0.0);\n";
        else out += "    const HDual x" + std::to_string(i) + "(in[" + std::to_string(i) + "]);\n";
    }
    for (int r = 0; r < n_regs; r++) out += "    HDual r" + std::to_string(r) + ";\n";
    out += "    HDual out(0.0);\n";
    auto V = [&](int idx) { return idx < n_in ? "x" + std::to_string(idx) : "r" + std::to_string(idx - n_in); };
    std::string ind = "    ";
    for (int k = 0; k < n_ops; k++) {
        const int32_t* op = &ops[5 * (size_t)k];
        const int type = op[0], dst = op[1], a = op[2], b = op[3], cond = op[4];
        if (type == OP_BRANCH) {
            if (cond == -2) {
                ind.resize(ind.size() - 4);
                out += ind + "}\n";
            } else if (a == 0) {
                out += ind + "if (" + V(cond) + ".v > 0.0) {\n";
                ind += "    ";
            } else {
                out += ind.substr(4) + "} else {\n";
            }
            continue;
        }
        if (type == OP_SYMBOL) {
            out += ind + "out = " + V(a) + ";\n";
            continue;
        }
        std::string e;
        switch (type) {
            case OP_ZERO: e = "HDual(0.0)"; break;
            case OP_ONE: e = "HDual(1.0)"; break;
            case OP_CONST: e = "HDual(" + rtc_double(consts[(size_t)k]) + ")"; break;
            case OP_ADD: e = V(a) + " + " + V(b); break;
            case OP_SUB: e = V(a) + " - " + V(b); break;
            case OP_MUL: e = V(a) + " * " + V(b); break;
            case OP_RECIP: e = "inv(" + V(a) + ")"; break;
            case OP_POWN: e = "cop_pown(" + V(a) + ", " + std::to_string(b) + ")"; break;
            case OP_POWF: e = "cop_powf(" + V(a) + ", " + V(b) + ")"; break;
            case OP_SQRT: e = "sqrt(" + V(a) + ")"; break;
            case OP_LN: e = "cop_ln(" + V(a) + ")"; break;
            case OP_LOG10: e = "cop_log10(" + V(a) + ")"; break;
            case OP_EXP: e = "cop_exp(" + V(a) + ")"; break;
            case OP_SIN: e = "sin(" + V(a) + ")"; break;
            case OP_COS: e = "cos(" + V(a) + ")"; break;
            case OP_TAN: e = "cop_tan(" + V(a) + ")"; break;
            case OP_ASIN: e = "cop_asin(" + V(a) + ")"; break;
            case OP_ACOS: e = "acos(" + V(a) + ")"; break;
            case OP_ATAN: e = "atan(" + V(a) + ")"; break;
            default: e = "HDual(0.0)"; break;  // Print
        }
        out += ind + V(dst) + " = " + e + ";\n";
    }
    out += "    return out;\n}\n";
}
std::string emit_source(const CustomProgram& G, const std::vector<int32_t>& in_dof, int NB)
{
    std::string s;
    s.reserve(1 << 16);
    s += "// emitted by libmistark (custom.hip) for hipRTC\n";
    s += RTC_POTARGS_SRC;
    s += RTC_HDUAL_SRC;
    s += RTC_MATH_SRC;
    s += "using namespace mistark;\n";
    emit_program(s, "prog_energy", G.ops, G.consts, G.n_in, G.n_regs, in_dof);
    const bool has_cond = !G.cconsts.empty();
    if (has_cond) emit_program(s, "prog_condition", G.cops, G.cconsts, G.n_in, G.n_cregs, in_dof);
    const int n = 3 * NB;
    for (int MODE = 0; MODE < 3; MODE++) {
        const int NP = MODE == 0 ? 1 : (MODE == 1 ? n : n * (n + 1) / 2);
        s += "extern \"C\" __global__ __launch_bounds__(256) void mistark_custom_k" + std::to_string(MODE) +
             "(PotArgs a, const double* __restrict__ sum_data, int sum_n, double* __restrict__ elemE, double* __restrict__ elemH, double* __restrict__ grad)\n{\n";
        s += "    constexpr int NB = " + std::to_string(NB) + ", n = 3 * NB, NP = " + std::to_string(NP) + ";\n";
        s += "    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;\n"
             "    if (t >= (long long)a.e_count * NP) return;\n"
             "    const int le = (int)(t / NP);\n"
             "    const int e = a.elem_list ? (int)a.elem_list[le] : a.e_begin + le;\n"
             "    const int pe = a.elem_list ? le : e;\n"
             "    int rem = (int)(t - (long long)le * NP);\n"
             "    const bool first = rem == 0;\n"
             "    int i = -1, j = -1;\n";
        if (MODE == 1) s += "    i = j = rem;\n";
        if (MODE == 2) s += "    i = 0;\n    while (rem >= n - i) { rem -= n - i; i++; }\n    j = i + rem;\n";
        s += "    (void)n; (void)first;\n    const int32_t* ce = a.conn + (size_t)e * a.conn_stride;\n";
        s += "    double in[" + std::to_string(std::max(G.n_in, 1)) + "];\n";
        int o = 0;
        for (size_t b = 0; b < G.strides.size(); b++) {
            const int S = G.strides[b];
            s += "    {\n        const int col = a.conn_col[" + std::to_string(b) + "];\n        const double* src = a.arr[" + std::to_string(b) + "] + (col < 0 ? 0 : (size_t)ce[col]) * " +
                 std::to_string(S) + ";\n";
            for (int c = 0; c < S; c++) s += "        in[" + std::to_string(o + c) + "] = src[" + std::to_string(c) + "];\n";
            s += "    }\n";
            o += S;
        }
        auto sum_set = [&](const char* ind) {
            std::string r;
            for (int c = 0; c < G.sum_stride; c++)
                r += std::string(ind) + "in[" + std::to_string(G.sum_first + c) + "] = sum_data[(size_t)it * " + std::to_string(G.sum_stride) + " + " + std::to_string(c) + "];\n";
            return r;
        };
        s += "    bool on = true;\n";
        if (has_cond) {
            if (G.sum_n == 0) s += "    on = prog_condition(in, -1, -1).v > 0.0;\n";
            else s += "    {\n        double cv = 0.0;\n        for (int it = 0; it < sum_n; it++) {\n" + sum_set("            ") + "            cv += prog_condition(in, -1, -1).v;\n        }\n        on = cv > 0.0;\n    }\n";
        }
        s += "    HDual r(0.0);\n    if (on) {\n";
        if (G.sum_n == 0) s += "        r = prog_energy(in, i, j);\n";
        else s += "        for (int it = 0; it < sum_n; it++) {\n" + sum_set("            ") + "            r = r + prog_energy(in, i, j);\n        }\n";
        s += "    }\n";
        if (MODE == 2)
            s += "    {\n        const int ba = i / 3, ii = i - 3 * ba, bb = j / 3, jj = j - 3 * bb;\n"
                 "        elemH[((size_t)(ba * NB + bb) * a.n_pool + pe) * 9 + ii * 3 + jj] = r.ab;\n"
                 "        elemH[((size_t)(bb * NB + ba) * a.n_pool + pe) * 9 + jj * 3 + ii] = r.ab;\n    }\n";
        if (MODE >= 1)
            s += "    if (i == j && on) {\n        const int ba = i / 3, ii = i - 3 * ba;\n"
                 "        const int node = a.conn[(size_t)e * a.conn_stride + a.dof_col[ba]];\n"
                 "        if (a.hot_base[ba] >= 0) atomicAdd(&a.grad_hot[((size_t)(blockIdx.x & (HOT_WAYS - 1)) * a.n_hot + a.hot_base[ba] + node) * 3 + ii], r.a);\n"
                 "        else atomicAdd(&grad[3 * (size_t)(a.dof_row_off[ba] + node) + ii], r.a);\n    }\n";
        s += "    if (first) elemE[pe] = energy_here(a, e) ? r.v : 0.0;\n}\n";
    }
    return s;
}
double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) {
        h ^= ch;
        h *= 1099511628211ull;
    }
    return h;
}
// ---- disk cache of emitted kernels' code objects ----------------------------------------------------------------------------------------
// A code object found in the cache is loaded and RUN on the device, so the cache is only used when it is provably this user's own:
// the directory is created (0700) BEFORE any lookup and must then be a real directory (no symlink) owned by getuid() that neither group
// nor others can write; files are opened O_NOFOLLOW, must be regular files of this user, and carry a header that binds them to the
// source text (length + a second, independent 64-bit hash besides the one in the file name), the target architecture, the compile
// options and the hipRTC version. Anything else: the cache is skipped (one line on stderr) and the kernel is compiled.
uint64_t hash2(const std::string& s)  // (independent of fnv1a: another multiplier, the bytes taken in reverse, a final avalanche)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)s.size();
    for (size_t i = s.size(); i-- > 0;) {
        h ^= (unsigned char)s[i];
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 32);
}
// the architecture the emitted kernels are compiled for: the current device's when there is one, else the one this library was built for
std::string rtc_arch()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
        std::string a = prop.gcnArchName;  // "gfx950:sramecc+:xnack-" -> "gfx950"
        const size_t colon = a.find(':');
        return colon == std::string::npos ? a : a.substr(0, colon);
    }
    (void)hipGetLastError();
#ifdef MISTARK_ARCH
    return MISTARK_ARCH;
#else
    return "gfx950";
#endif
}
bool make_private_dir(const std::string& d)
{
    if (::mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) return false;
    struct stat st;
    if (::lstat(d.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// the cache directory, or "" when there is none that can be trusted
std::string rtc_cache_dir()
{
    std::string d;
    if (const char* e = std::getenv("MISTARK_RTC_CACHE")) {
        if (!e[0] || std::string(e) == "0" || std::string(e) == "off") return std::string();
        d = e;
    } else {
        std::string base;
        const char* x = std::getenv("XDG_CACHE_HOME");
        const char* h = std::getenv("HOME");
        if (x && x[0] == '/') base = x;
        else if (h && h[0] == '/') {
            base = std::string(h) + "/.cache";
            (void)::mkdir(base.c_str(), 0700);
        }
        d = !base.empty() ? base + "/mistark_rtc" : "/tmp/mistark_rtc_cache_" + std::to_string((long long)getuid());
    }
    if (!make_private_dir(d)) {
        static std::string warned;
        if (warned != d) {
            warned = d;
            std::fprintf(stderr, "mistark: hipRTC cache directory '%s' is not a directory of uid %ld closed to group / others: cache not used\n", d.c_str(), (long)getuid());
        }
        return std::string();
    }
    return d;
}
struct RtcCacheHeader
{
    char magic[8];  // "MISRTC02"
    uint64_t src_len, src_hash2, key_hash, code_len;
};
std::vector<char> rtc_cache_load(const std::string& path, const std::string& src, uint64_t key_hash)
{
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return {};
    std::vector<char> code;
    struct stat st;
    RtcCacheHeader h;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && st.st_size > (off_t)sizeof(h) &&
        ::read(fd, &h, sizeof(h)) == (ssize_t)sizeof(h) && std::memcmp(h.magic, "MISRTC02", 8) == 0 && h.src_len == src.size() && h.src_hash2 == hash2(src) &&
        h.key_hash == key_hash && h.code_len > 64 && h.code_len == (uint64_t)st.st_size - sizeof(h)) {
        code.resize(h.code_len);
        size_t got = 0;
        while (got < code.size()) {
            const ssize_t n = ::read(fd, code.data() + got, code.size() - got);
            if (n <= 0) break;
            got += (size_t)n;
        }
        if (got != code.size() || std::memcmp(code.data(), "\x7f" "ELF", 4) != 0) code.clear();
    }
    ::close(fd);
    return code;
}
void rtc_cache_store(const std::string& path, const std::string& src, uint64_t key_hash, const std::vector<char>& code)
{  // best effort: written under a temporary name of this process (O_EXCL, 0600) and renamed
    const std::string tmp = path + "." + std::to_string((long long)getpid());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    RtcCacheHeader h;
    std::memcpy(h.magic, "MISRTC02", 8);
    h.src_len = src.size();
    h.src_hash2 = hash2(src);
    h.key_hash = key_hash;
    h.code_len = code.size();
    bool ok = ::write(fd, &h, sizeof(h)) == (ssize_t)sizeof(h);
    size_t put = 0;
    while (ok && put < code.size()) {
        const ssize_t n = ::write(fd, code.data() + put, code.size() - put);
        if (n <= 0) ok = false;
        else put += (size_t)n;
    }
    ok = ::close(fd) == 0 && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)::unlink(tmp.c_str());
}
// compiled code object of `src` (from the disk cache, or built now); empty + why on failure
std::vector<char> rtc_build(const std::string& src, std::string& why)
{
    HipRtc& R = hiprtc();
    const std::string arch = rtc_arch();
    const std::string arch_opt = "--offload-arch=" + arch;
    const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-pragma-once-outside-header"};
    constexpr int n_opts = 5;
    // what besides the source decides the code object: target, options, compiler version
    std::string key = arch;
    for (const char* o : opts) key += std::string("|") + o;
    int vmaj = 0, vmin = 0;
    if (R.lib && R.Version) (void)R.Version(&vmaj, &vmin);
    key += "|hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin);
    const uint64_t key_hash = fnv1a(key);
    const std::string dir = rtc_cache_dir();
    std::string path;
    if (!dir.empty()) {
        char name[96];
        std::snprintf(name, sizeof(name), "/%016llx_%016llx_%s.hsaco", (unsigned long long)fnv1a(src), (unsigned long long)key_hash, arch.c_str());
        path = dir + name;
        std::vector<char> code = rtc_cache_load(path, src, key_hash);
        if (!code.empty()) return code;
    }
    if (!R.lib) {
        why = "libhiprtc.so not available";
        return {};
    }
    void* prog = nullptr;
    if (R.CreateProgram(&prog, src.c_str(), "mistark_custom.hip", 0, nullptr, nullptr) != 0) {
        why = "hiprtcCreateProgram failed";
        return {};
    }
    const int rc = R.CompileProgram(prog, n_opts, opts);
    std::vector<char> code;
    if (rc != 0) {
        size_t n = 0;
        R.GetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) R.GetProgramLog(prog, &log[0]);
        why = "hipRTC compile failed (" + arch + "): " + log.substr(0, 2000);
    } else {
        size_t n = 0;
        R.GetCodeSize(prog, &n);
        code.resize(n);
        R.GetCode(prog, code.data());
    }
    R.DestroyProgram(&prog);
    if (!code.empty() && !path.empty()) rtc_cache_store(path, src, key_hash, code);
    return code;
}
int rtc_max_ops()
{
    static const int v = [] {
        const char* e = std::getenv("MISTARK_RTC_MAX_OPS");
        return e ? std::atoi(e) : 3000;
    }();
    return v;
}
}  // namespace
// the emitted kernels of P's program for the bindings as they are now; nullptr: use the interpreter
RtcKernels* rtc_kernels(Context& c, Potential& P, const std::vector<int32_t>& in_dof)
{
    CustomProgram& G = *P.prog;
    if (!c.custom_rtc || G.rtc_failed) return nullptr;
    if ((int)(G.consts.size() + G.cconsts.size()) > rtc_max_ops()) return nullptr;  // (hyper-dual straight-line code of 10^4 ops compiles for minutes: interpreted)
    std::string key = std::to_string(P.NB) + "|" + std::to_string(G.sum_first) + "," + std::to_string(G.sum_stride) + "," + (G.sum_n > 0 ? "s" : "-") + "|";
    for (int32_t d : in_dof) key += std::to_string(d) + ",";
    if (G.rtc && G.rtc_key == key) return G.rtc.get();
    const double t0 = now_seconds();
    std::string why;
    const std::string src = emit_source(G, in_dof, P.NB);
    if (const char* dump = std::getenv("MISTARK_RTC_DUMP")) {
        if (FILE* f = std::fopen((std::string(dump) + "/" + P.name + ".hip").c_str(), "w")) {
            std::fputs(src.c_str(), f);
            std::fclose(f);
        }
    }
    const std::vector<char> code = rtc_build(src, why);
    auto k = std::make_shared<RtcKernels>();
    if (!code.empty()) {
        if (hipModuleLoadData(&k->mod, code.data()) != hipSuccess) why = "hipModuleLoadData failed";
        else
            for (int m = 0; m < 3 && why.empty(); m++)
                if (hipModuleGetFunction(&k->fn[m], k->mod, ("mistark_custom_k" + std::to_string(m)).c_str()) != hipSuccess) why = "emitted kernel not found in the module";
    }
    if (!why.empty() || code.empty()) {
        (void)hipGetLastError();
        G.rtc_failed = true;
        std::fprintf(stderr, "mistark: user-defined potential '%s': no emitted kernel (%s); the device interpreter runs it\n", P.name.c_str(), why.c_str());
        return nullptr;
    }
    G.rtc = k;
    G.rtc_key = key;
    c.n_rtc_builds++;
    c.t_rtc_builds += now_seconds() - t0;
    return k.get();
}

// What the emitter writes for an op sequence, and whether hipRTC compiles it — no context, no GPU (the CPU-side test of the emitter;
// mistark_custom_emit in api.cpp). in_dof: per input the local DoF component it seeds, or -1.
std::string custom_emit_source(const std::string& name, const int32_t* strides, int n_bindings, const int32_t* in_dof, const int32_t* ops, const double* consts, int n_ops, int n_inputs,
                               const int32_t* cond_ops, const double* cond_consts, int n_cond_ops, int NB, bool compile, size_t* code_bytes)
{
    std::shared_ptr<CustomProgram> G = make_custom_program(name, strides, n_bindings, ops, consts, n_ops, n_inputs, cond_ops, cond_consts, n_cond_ops);
    const std::vector<int32_t> dof(in_dof, in_dof + n_inputs);
    const std::string src = emit_source(*G, dof, NB);
    if (code_bytes) *code_bytes = 0;
    if (compile) {
        std::string why;
        const std::vector<char> code = rtc_build(src, why);
        if (code.empty()) throw Error("custom potential '" + name + "': " + why);
        if (code_bytes) *code_bytes = code.size();
    }
    return src;
}

// Evaluation of a custom potential (called by kernels.hip: launch_eval_kind for P.kind == KIND_CUSTOM)
static void launch_eval_custom_impl(Context& c, Potential& P, int mode);
void launch_eval_custom(Context& c, Potential& P, int mode)
{
    if (!c.custom_timing) return launch_eval_custom_impl(c, P, mode);
    // option custom_timing (measurement): HIP events around the potential's own launch, synchronised; counter "custom_kernel_us"
    struct Events  // (destroyed on every way out: launch_eval_custom_impl and MS_CHECK throw)
    {
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events()
        {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
    } ev;
    MS_CHECK(hipEventCreate(&ev.e[0]));
    MS_CHECK(hipEventCreate(&ev.e[1]));
    MS_CHECK(hipEventRecord(ev.e[0], c.stream));
    launch_eval_custom_impl(c, P, mode);
    MS_CHECK(hipEventRecord(ev.e[1], c.stream));
    MS_CHECK(hipEventSynchronize(ev.e[1]));
    float ms = 0.f;
    MS_CHECK(hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
    c.custom_kernel_us += 1e3 * (double)ms;
}
static void launch_eval_custom_impl(Context& c, Potential& P, int mode)
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
    if (RtcKernels* K = rtc_kernels(c, P, in_dof)) {  // the kernels emitted for this op sequence (hipRTC)
        const int m = mode == MISTARK_EVAL_P ? 0 : (mode == MISTARK_EVAL_P_G ? 1 : 2);
        const long long threads = (long long)P.args.e_count * (m == 0 ? 1 : (m == 1 ? n : n * (n + 1) / 2));
        if (threads <= 0) return;
        PotArgs a = P.args;
        const double* sum_data = G.d_sum.p;
        int sum_n = G.sum_n;
        double* H = m == 2 ? c.elemH.p + P.h_off : nullptr;
        double* g = m >= 1 ? c.grad.p : nullptr;
        void* params[] = {&a, &sum_data, &sum_n, &E, &H, &g};
        MS_CHECK(hipModuleLaunchKernel(K->fn[m], grid(threads).x, 1, 1, 256, 1, 1, 0, c.stream, params, nullptr));
        c.n_rtc_launches++;
        return;
    }
    if (mode == MISTARK_EVAL_P)
        hipLaunchKernelGGL(k_eval_custom<0>, grid(P.args.e_count), dim3(256), 0, c.stream, P.args, pd, E, (double*)nullptr, (double*)nullptr);
    else if (mode == MISTARK_EVAL_P_G)
        hipLaunchKernelGGL(k_eval_custom<1>, grid((long long)P.args.e_count * n), dim3(256), 0, c.stream, P.args, pd, E, (double*)nullptr, c.grad.p);
    else
        hipLaunchKernelGGL(k_eval_custom<2>, grid((long long)P.args.e_count * (n * (n + 1) / 2)), dim3(256), 0, c.stream, P.args, pd, E, c.elemH.p + P.h_off, c.grad.p);
}

}  // namespace mistark
