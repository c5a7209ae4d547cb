// api.cpp — extern "C" entry points of libmistark.so (see include/mistark.h for the contract and reference citations).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "dist.hpp"
#include "engine.hpp"

using namespace mistark;

#define API_BEGIN        \
    if (!ctx) return -1; \
    int _ret = 0;        \
    (void)_ret;          \
    ::mistark::DryScope _dry(ctx->c.dry); \
    try {                \
        if (!ctx->c.dry) (void)hipSetDevice(ctx->c.device); /* the current device is per host thread: contexts may be driven from any thread */
#define API_END(ret)                   \
    }                                  \
    catch (const std::exception& e)    \
    {                                  \
        ctx->c.last_error = e.what();  \
        return -1;                     \
    }                                  \
    return ret;

namespace mistark {
int register_potential(Context& c, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings)
{
    const int kind = find_kind(name);
    if (kind < 0) throw Error(std::string("unknown potential '") + name + "' (no MI355X kernel registered under this name)");
    if (n_bindings != kind_nbind(kind)) throw Error(std::string("potential '") + name + "': expected " + std::to_string(kind_nbind(kind)) + " bindings, got " + std::to_string(n_bindings));
    int strides[MAX_BIND];
    kind_strides(kind, strides);
    for (int b = 0; b < n_bindings; b++) {
        if (bindings[b].array < 0 || bindings[b].array >= (int)c.arrays.size()) throw Error(std::string("potential '") + name + "': bad array id in binding " + std::to_string(b));
        if (bindings[b].stride != strides[b] || c.arrays[bindings[b].array].stride != strides[b])
            throw Error(std::string("potential '") + name + "': binding " + std::to_string(b) + " must have stride " + std::to_string(strides[b]));
        if (bindings[b].conn_col >= conn_stride) throw Error(std::string("potential '") + name + "': connectivity column out of range");
    }
    if (n_elem < 0 || conn_stride <= 0) throw Error("bad connectivity shape");
    Potential* P = nullptr;
    int id = -1;
    for (size_t i = 0; i < c.pots.size(); i++)
        if (c.pots[i].name == name) {
            P = &c.pots[i];
            id = (int)i;
        }
    if (!P) {
        c.pots.emplace_back();
        P = &c.pots.back();
        id = (int)c.pots.size() - 1;
    }
    P->name = name;
    P->kind = kind;
    P->NB = kind_nb(kind);
    P->n_elem = n_elem;
    P->conn_stride = conn_stride;
    if (conn) P->conn_host.assign(conn, conn + (size_t)n_elem * conn_stride);
    else P->conn_host.clear();
    P->bindings.assign(bindings, bindings + n_bindings);
    P->conn_dirty = true;
    c.layout_dirty = true;
    return id;
}
int register_custom_potential(Context& c, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings,
                              const int32_t* ops, const double* consts, int32_t n_ops, int32_t n_inputs, const int32_t* cond_ops, const double* cond_consts, int32_t n_cond_ops)
{
    if (!name || !*name) throw Error("custom potential: empty name");
    if (n_bindings <= 0 || n_bindings > MAX_BIND) throw Error(std::string("custom potential '") + name + "': between 1 and " + std::to_string(MAX_BIND) + " bindings");
    if (n_elem < 0 || conn_stride <= 0) throw Error("bad connectivity shape");
    std::vector<int32_t> strides((size_t)n_bindings);
    int nb = 0;
    for (int b = 0; b < n_bindings; b++) {
        if (bindings[b].array < 0 || bindings[b].array >= (int)c.arrays.size()) throw Error(std::string("custom potential '") + name + "': bad array id in binding " + std::to_string(b));
        const Array& arr = c.arrays[bindings[b].array];
        if (bindings[b].stride != arr.stride) throw Error(std::string("custom potential '") + name + "': binding " + std::to_string(b) + " stride differs from its array's");
        if (bindings[b].conn_col >= conn_stride) throw Error(std::string("custom potential '") + name + "': connectivity column out of range");
        if (arr.dof_set >= 0) {
            if (arr.stride != 3 || bindings[b].conn_col < 0) throw Error(std::string("custom potential '") + name + "': DoF bindings are 3-vectors fetched through a connectivity column");
            nb++;
        }
        strides[b] = bindings[b].stride;
    }
    if (nb == 0 || nb > MAX_NB) throw Error(std::string("custom potential '") + name + "': between 1 and " + std::to_string(MAX_NB) + " DoF bindings");
    auto prog = make_custom_program(name, strides.data(), n_bindings, ops, consts, n_ops, n_inputs, cond_ops, cond_consts, n_cond_ops);
    Potential* P = nullptr;
    int id = -1;
    for (size_t i = 0; i < c.pots.size(); i++)
        if (c.pots[i].name == name) {
            P = &c.pots[i];
            id = (int)i;
        }
    if (!P) {
        c.pots.emplace_back();
        P = &c.pots.back();
        id = (int)c.pots.size() - 1;
    }
    P->name = name;
    P->kind = KIND_CUSTOM;
    P->prog = prog;
    P->NB = nb;
    P->n_elem = n_elem;
    P->conn_stride = conn_stride;
    if (conn) P->conn_host.assign(conn, conn + (size_t)n_elem * conn_stride);
    else P->conn_host.clear();
    P->bindings.assign(bindings, bindings + n_bindings);
    P->conn_dirty = true;
    c.layout_dirty = true;
    return id;
}
}  // namespace mistark

extern "C" {

const char* mistark_version(void) { return "mistark 0.1 (gfx950, HIP)"; }

int mistark_create(int device, mistark_ctx** out)
{
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -2;  // no GPU: the product path fails loudly, there is no CPU fallback
    if (device < 0 || device >= n) return -3;
    // the Newton loop synchronises with the device a few dozen times per iteration (scalars only): spin instead of sleeping
    (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    if (hipSetDevice(device) != hipSuccess) return -4;
    mistark_ctx* ctx = new mistark_ctx();
    ctx->c.device = device;
    if (hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return -5;
    }
    *out = ctx;
    // MISTARK_OPTIONS="name=value,name=value": the mistark_set_option switches for a process that cannot be edited (a test suite, a profiler run)
    if (const char* env = std::getenv("MISTARK_OPTIONS")) {
        std::string all(env);
        size_t at = 0;
        while (at < all.size()) {
            const size_t end = std::min(all.find(',', at), all.size());
            const std::string item = all.substr(at, end - at);
            const size_t eq = item.find('=');
            if (eq == std::string::npos || mistark_set_option(ctx, item.substr(0, eq).c_str(), std::atoi(item.c_str() + eq + 1)) != 0) {
                std::fprintf(stderr, "mistark: bad entry '%s' in MISTARK_OPTIONS\n", item.c_str());
                mistark_destroy(ctx);
                *out = nullptr;
                return -6;
            }
            at = end + 1;
        }
    }
    return 0;
}
int mistark_create_dry(mistark_ctx** out)
{
    if (!out) return -1;
    mistark_ctx* ctx = new mistark_ctx();  // (Context::dry reaches DevBuf::ensure through the DryScope of every entry point)
    ctx->c.dry = true;
    *out = ctx;
    return 0;
}
void mistark_destroy(mistark_ctx* ctx)
{
    if (!ctx) return;
    if (!ctx->c.dry) {
        (void)hipSetDevice(ctx->c.device);
        (void)hipStreamSynchronize(ctx->c.stream);
    }
    delete ctx;
}
// What has been registered, as JSON: DoF sets, arrays, potentials with their bindings. Arrays are named by the DoF set they view or by the
// order in which they were first bound ("a0", "a1", ...): two registrations of the same scene by different callers compare equal.
int64_t mistark_describe(mistark_ctx* ctx, char* buf, int64_t cap)
{
    if (!ctx) return -1;
    Context& c = ctx->c;
    auto esc = [](const std::string& in) {  // JSON string escaping (labels and names are the caller's)
        std::string out;
        for (unsigned char ch : in) {
            if (ch == '"' || ch == '\\') {
                out += '\\';
                out += (char)ch;
            } else if (ch < 0x20) {
                char b[8];
                std::snprintf(b, sizeof b, "\\u%04x", ch);
                out += b;
            } else {
                out += (char)ch;
            }
        }
        return out;
    };
    std::string o = "{\"dof_sets\":[";
    for (size_t i = 0; i < c.dof_sets.size(); i++) o += std::string(i ? "," : "") + "{\"label\":\"" + esc(c.dof_sets[i].label) + "\",\"n\":" + std::to_string(c.dof_sets[i].n) + "}";
    o += "],\"potentials\":[";
    std::vector<int> alias(c.arrays.size(), -1);
    int n_alias = 0;
    auto name_of = [&](int a) {
        const Array& A = c.arrays[(size_t)a];
        int set = A.dof_set;
        if (set < 0)
            for (size_t k = 0; k < c.dof_sets.size(); k++)
                if (A.host && A.host == c.dof_sets[k].host) set = (int)k;
        if (set >= 0) return "dof:" + esc(c.dof_sets[(size_t)set].label);
        if (alias[(size_t)a] < 0) alias[(size_t)a] = n_alias++;
        return "a" + std::to_string(alias[(size_t)a]);
    };
    for (size_t i = 0; i < c.pots.size(); i++) {
        const Potential& P = c.pots[i];
        o += std::string(i ? "," : "") + "{\"name\":\"" + esc(P.name) + "\",\"conn_stride\":" + std::to_string(P.conn_stride) + ",\"n_elem\":" + std::to_string(P.n_elem) +
             ",\"dynamic\":" + std::to_string(P.part) + ",\"bindings\":[";
        for (size_t b = 0; b < P.bindings.size(); b++) {
            const mistark_binding& B = P.bindings[b];
            o += std::string(b ? "," : "") + "[\"" + name_of(B.array) + "\"," + std::to_string(B.stride) + "," + std::to_string(B.conn_col) + "," +
                 std::to_string(c.arrays[(size_t)B.array].n_items) + "]";
        }
        o += "]}";
    }
    o += "]}";
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(o.size(), (size_t)cap - 1);
        std::memcpy(buf, o.data(), n);
        buf[n] = 0;
    }
    return (int64_t)o.size() + 1;
}
const char* mistark_last_error(mistark_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : "null context"; }

int mistark_n_supported_potentials(void) { return n_kinds(); }
const char* mistark_supported_potential(int i) { return (i >= 0 && i < n_kinds()) ? kind_name(i) : nullptr; }

int mistark_add_dof_set(mistark_ctx* ctx, const char* label, double* host, int64_t n_scalars)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    if (n_scalars < 0 || n_scalars % 3 != 0) throw Error("DoF set size must be a non-negative multiple of 3");
    if (n_scalars > 0 && !host) throw Error("DoF set without a host array");
    DofSet s;
    s.label = label ? label : "";
    s.host = host;
    s.n = n_scalars;
    ctx->c.dof_sets.push_back(s);
    ctx->c.layout_dirty = true;
    _ret = (int)ctx->c.dof_sets.size() - 1;
    API_END(_ret)
}
int mistark_resize_dof_set(mistark_ctx* ctx, int set, double* host, int64_t n_scalars)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    if (set < 0 || set >= (int)ctx->c.dof_sets.size()) throw Error("bad DoF set");
    if (n_scalars < 0 || n_scalars % 3 != 0) throw Error("DoF set size must be a non-negative multiple of 3");
    if (n_scalars > 0 && !host) throw Error("DoF set without a host array");
    Context& c = ctx->c;
    const double* old = c.dof_sets[set].host;
    if (old != host || c.dof_sets[set].n != n_scalars) host_range_unpin(c, old);
    c.dof_sets[set].host = host;
    c.dof_sets[set].n = n_scalars;
    for (auto& a : c.arrays)
        if (a.dof_set == set) {
            a.host = host;
            a.n_items = n_scalars / a.stride;
        }
    (void)old;
    c.ndofs = -1;  // force re-upload of the DoF vector
    c.layout_dirty = true;
    API_END(0)
}

int mistark_array(mistark_ctx* ctx, const double* host, int64_t n_items, int stride)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    if (stride <= 0) throw Error("bad stride");
    if (n_items < 0) throw Error("negative item count");
    for (size_t i = 0; i < c.arrays.size(); i++) {
        if (host != nullptr && c.arrays[i].host == host && c.arrays[i].stride == stride) {
            if (c.arrays[i].n_items != n_items) {
                c.arrays[i].n_items = n_items;
                c.arrays[i].need_upload = true;
                c.layout_dirty = true;
            }
            return (int)i;
        }
    }
    c.arrays.emplace_back();
    Array& a = c.arrays.back();
    a.host = host;
    a.n_items = n_items;
    a.stride = stride;
    for (size_t s = 0; s < c.dof_sets.size(); s++)
        if (host != nullptr && (const double*)c.dof_sets[s].host == host) {  // (empty containers all have the null address: mistark_dof_array)
            a.dof_set = (int)s;
            if (c.dof_sets[s].n != n_items * stride) throw Error("array bound on a DoF set must cover the whole set");
        }
    c.layout_dirty = true;
    _ret = (int)c.arrays.size() - 1;
    API_END(_ret)
}
int mistark_dof_array(mistark_ctx* ctx, int set, int stride)
{
    API_BEGIN
    Context& c = ctx->c;
    if (set < 0 || set >= (int)c.dof_sets.size()) throw Error("bad DoF set");
    if (stride <= 0 || c.dof_sets[set].n % stride != 0) throw Error("bad stride for a view of this DoF set");
    for (size_t i = 0; i < c.arrays.size(); i++)
        if (c.arrays[i].dof_set == set && c.arrays[i].stride == stride) return (int)i;
    c.arrays.emplace_back();
    Array& a = c.arrays.back();
    a.host = c.dof_sets[set].host;
    a.n_items = c.dof_sets[set].n / stride;
    a.stride = stride;
    a.dof_set = set;
    c.layout_dirty = true;
    _ret = (int)c.arrays.size() - 1;
    API_END(_ret)
}
int mistark_array_rebind(mistark_ctx* ctx, int array, const double* host, int64_t n_items)
{
    API_BEGIN
    ctx->c.touch_array(array);
    Context& c = ctx->c;
    if (array < 0 || array >= (int)c.arrays.size()) throw Error("bad array id");
    Array& a = c.arrays[array];
    if (a.dof_set >= 0) throw Error("use mistark_resize_dof_set for DoF arrays");
    if (a.host != host || a.n_items != n_items) {
        bool shared = false;  // (another array — the same container bound at another stride — may still be on the old range)
        for (const Array& o : c.arrays) shared = shared || (&o != &a && o.host == a.host);
        if (!shared) host_range_unpin(c, a.host);
    }
    a.host = host;
    a.n_items = n_items;
    a.need_upload = true;
    c.layout_dirty = true;
    API_END(0)
}
static void upload_one(Context& c, Array& a)
{
    if (a.dof_set >= 0) {
        const DofSet& s = c.dof_sets[a.dof_set];
        if (s.n > 0) MS_CHECK(hipMemcpyAsync(c.u.p + s.offset, s.host, s.n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    } else {
        const size_t n = (size_t)a.n_items * a.stride;
        if (n > 0 && a.host) h2d_staged(c, a.dev, a.host, n * sizeof(double));
    }
    a.need_upload = false;
}
int mistark_upload(mistark_ctx* ctx, int array)
{
    API_BEGIN
    ctx->c.touch_array(array);
    Context& c = ctx->c;
    if (c.layout_dirty) {
        // sizes may have changed: mark and let prepare() do the copy
        if (array < 0) for (auto& a : c.arrays) a.need_upload = true;
        else if (array < (int)c.arrays.size()) c.arrays[array].need_upload = true;
        else throw Error("bad array id");
        prepare(c);
        if (array < 0) for (auto& a : c.arrays) upload_one(c, a);
        else upload_one(c, c.arrays[array]);
    } else {
        if (array < 0) for (auto& a : c.arrays) upload_one(c, a);
        else if (array < (int)c.arrays.size()) upload_one(c, c.arrays[array]);
        else throw Error("bad array id");
    }
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
static void download_one(Context& c, Array& a)
{
    const size_t n = (size_t)a.n_items * a.stride;
    if (n > 0 && a.host) MS_CHECK(hipMemcpyAsync(const_cast<double*>(a.host), a.dev, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
}
int mistark_download(mistark_ctx* ctx, int array)
{
    API_BEGIN
    Context& c = ctx->c;
    prepare(c);
    if (array < 0) for (auto& a : c.arrays) download_one(c, a);
    else if (array < (int)c.arrays.size()) download_one(c, c.arrays[array]);
    else throw Error("bad array id");
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
int mistark_array_axpby(mistark_ctx* ctx, int dst, double a, int x, double b, int y)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    prepare(c);
    const int na = (int)c.arrays.size();
    if (dst < 0 || dst >= na || x < 0 || x >= na || y >= na) throw Error("bad array id");
    const int64_t n = c.arrays[dst].n_items * c.arrays[dst].stride;
    if (c.arrays[x].n_items * c.arrays[x].stride != n || (y >= 0 && c.arrays[y].n_items * c.arrays[y].stride != n)) throw Error("axpby: size mismatch");
    vec_axpby(c, c.arrays[dst].dev, a, c.arrays[x].dev, b, y >= 0 ? c.arrays[y].dev : nullptr, n);
    API_END(0)
}
int mistark_array_fill(mistark_ctx* ctx, int dst, double value)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    prepare(c);
    if (dst < 0 || dst >= (int)c.arrays.size()) throw Error("bad array id");
    vec_fill(c, c.arrays[dst].dev, value, c.arrays[dst].n_items * c.arrays[dst].stride);
    API_END(0)
}

int mistark_potential(mistark_ctx* ctx, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    _ret = register_potential(ctx->c, name, conn, n_elem, conn_stride, bindings, n_bindings);
    API_END(_ret)
}
int mistark_potential_custom(mistark_ctx* ctx, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings,
                             const int32_t* ops, const double* constants, int32_t n_ops, int32_t n_inputs, const int32_t* cond_ops, const double* cond_constants, int32_t n_cond_ops)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    _ret = register_custom_potential(ctx->c, name, conn, n_elem, conn_stride, bindings, n_bindings, ops, constants, n_ops, n_inputs, cond_ops, cond_constants, n_cond_ops);
    API_END(_ret)
}
int mistark_potential_custom_set_summation(mistark_ctx* ctx, int potential, int32_t first_input, int32_t stride, int32_t n_iterations, const double* data)
{
    API_BEGIN
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    Potential& P = c.pots[(size_t)potential];
    if (!P.prog) throw Error("mistark_potential_custom_set_summation: '" + P.name + "' is not a custom potential");
    custom_program_set_summation(*P.prog, P.name, first_input, stride, n_iterations, data);
    c.touch();
    c.u_version++;
    API_END(0)
}
int mistark_potential_table(mistark_ctx* ctx, int potential, int32_t* conn, int64_t* n_elem, int32_t* conn_stride)
{
    API_BEGIN
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    const Potential& P = c.pots[(size_t)potential];
    if (P.conn_ext) throw Error("mistark_potential_table: the table of '" + P.name + "' lives on the device (mistark_contact_get_table)");
    if (n_elem) *n_elem = P.n_elem;
    if (conn_stride) *conn_stride = P.conn_stride;
    if (conn && !P.conn_host.empty()) std::memcpy(conn, P.conn_host.data(), std::min(P.conn_host.size(), (size_t)P.n_elem * P.conn_stride) * sizeof(int32_t));
    API_END(0)
}
int mistark_potential_binding_data(mistark_ctx* ctx, int potential, int binding, const double** host, int64_t* n_items, int32_t* stride)
{
    API_BEGIN
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    const Potential& P = c.pots[(size_t)potential];
    if (binding < 0 || binding >= (int)P.bindings.size()) throw Error("bad binding index");
    const Array& A = c.arrays[(size_t)P.bindings[(size_t)binding].array];
    if (host) *host = A.host;
    if (n_items) *n_items = A.n_items;
    if (stride) *stride = A.stride;
    API_END(0)
}
int mistark_find_potential(mistark_ctx* ctx, const char* name)
{
    if (!ctx || !name) return -1;
    for (size_t i = 0; i < ctx->c.pots.size(); i++)
        if (ctx->c.pots[i].name == name) return (int)i;
    return -1;
}
int mistark_potential_set_dynamic(mistark_ctx* ctx, int potential, int dynamic)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    const int part = dynamic ? 1 : 0;
    if (c.pots[potential].part != part) {
        c.pots[potential].part = part;
        c.layout_dirty = true;
        c.part[0].dirty = c.part[1].dirty = true;
    }
    API_END(0)
}
int mistark_potential_update_connectivity(mistark_ctx* ctx, int potential, const int32_t* conn, int32_t n_elem)
{
    API_BEGIN
    ctx->c.touch_potential(potential);
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    if (n_elem < 0) throw Error("bad connectivity shape");
    Potential& P = c.pots[potential];
    P.n_elem = n_elem;
    P.conn_host.assign(conn, conn + (size_t)n_elem * P.conn_stride);
    P.conn_dirty = true;   // -> only this potential's matrix part is re-patterned
    c.layout_dirty = true;
    API_END(0)
}

int64_t mistark_ndofs(mistark_ctx* ctx)
{
    if (!ctx) return -1;
    int64_t n = 0;
    for (auto& s : ctx->c.dof_sets) n += s.n;
    return n;
}
int mistark_get_dofs(mistark_ctx* ctx, double* u_host)
{
    API_BEGIN
    Context& c = ctx->c;
    prepare(c);
    MS_CHECK(hipMemcpyAsync(u_host, c.u.p, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
int mistark_set_dofs(mistark_ctx* ctx, const double* u_host)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    prepare(c);
    MS_CHECK(hipMemcpyAsync(c.u.p, u_host, (size_t)c.ndofs * sizeof(double), hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
int64_t mistark_custom_emit(const char* name, const int32_t* strides, int32_t n_bindings, const int32_t* in_dof, const int32_t* ops, const double* constants, int32_t n_ops,
                            int32_t n_inputs, const int32_t* cond_ops, const double* cond_constants, int32_t n_cond_ops, int32_t n_blocks, int32_t compile, char* out, int64_t cap)
{
    // (no context: errors go to `out`)
    try {
        if (!name || !strides || !in_dof || !ops || !constants) throw Error("mistark_custom_emit: null argument");
        size_t code = 0;
        const std::string src = custom_emit_source(name, strides, n_bindings, in_dof, ops, constants, n_ops, n_inputs, cond_ops, cond_constants, n_cond_ops, n_blocks, compile != 0, &code);
        if (out && cap > 0) std::snprintf(out, (size_t)cap, "%s", src.c_str());
        return compile ? (int64_t)code : (int64_t)src.size();
    } catch (const std::exception& e) {
        if (out && cap > 0) std::snprintf(out, (size_t)cap, "%s", e.what());
        return -1;
    }
}
int mistark_get_counter(mistark_ctx* ctx, const char* name, int64_t* out)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!name || !out) throw Error("mistark_get_counter: null argument");
    const std::string n = name;
    if (n == "proj_speculated") *out = c.n_proj_speculated;
    else if (n == "proj_adopted") *out = c.n_proj_adopted;
    else if (n == "multi_pgh_launches") *out = c.n_multi_pgh;
    else if (n == "dof_skips_verified") *out = c.n_dof_skips_verified;
    else if (n == "host_ranges_pinned") *out = c.n_pin_ok;
    else if (n == "host_ranges_not_pinned") *out = c.n_pin_failed;
    else if (n == "rtc_builds") *out = c.n_rtc_builds;
    else if (n == "custom_kernel_us") *out = (int64_t)c.custom_kernel_us;
    else if (n == "rtc_launches") *out = c.n_rtc_launches;
    else if (n == "rtc_build_ms") *out = (int64_t)(1e3 * c.t_rtc_builds);
    else if (n == "fused_solves") *out = c.n_fused_solves;
    else if (n == "unfused_solves") *out = c.n_unfused_solves;
    else if (n == "evt_tet_us") *out = (int64_t)c.evt_sum[1];
    else if (n == "evt_small_us") *out = (int64_t)c.evt_sum[2];
    else if (n == "evt_gather_us") *out = (int64_t)c.evt_sum[3];
    else if (n == "evt_main_us") *out = (int64_t)c.evt_sum[4];
    else if (n == "evt_pattern_us") *out = (int64_t)c.evt_sum[5];
    else if (n == "evt_n") *out = c.evt_n;
    else if (n == "eval_pgh_issue_us") *out = (int64_t)(1e6 * c.t_eval_issue);
    else if (n == "eval_pgh_wait_us") *out = (int64_t)(1e6 * c.t_eval_wait);
    else if (n == "contact_searches") *out = contact_searches(c, false);
    else if (n == "contact_repeated_searches") *out = contact_searches(c, true);
    else throw Error("mistark_get_counter: unknown counter '" + n + "'");
    API_END(0)
}
int mistark_dofs_to_host_arrays(mistark_ctx* ctx)
{
    API_BEGIN
    Context& c = ctx->c;
    prepare(c);
    for (auto& s : c.dof_sets)
        if (s.n > 0) {
            (void)host_range_pinned(c, s.host, (size_t)s.n * sizeof(double));
            MS_CHECK(hipMemcpyAsync(s.host, c.u.p + s.offset, s.n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        }
    MS_CHECK(hipStreamSynchronize(c.stream));
    c.u_host_version = c.u_version;
    API_END(0)
}
int mistark_dofs_to_host_arrays_if_changed(mistark_ctx* ctx)
{
    API_BEGIN
    Context& c = ctx->c;
    prepare(c);
    if (c.u_host_version != c.u_version) {
        for (auto& s : c.dof_sets)
            if (s.n > 0) {
                (void)host_range_pinned(c, s.host, (size_t)s.n * sizeof(double));
                MS_CHECK(hipMemcpyAsync(s.host, c.u.p + s.offset, s.n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
            }
        MS_CHECK(hipStreamSynchronize(c.stream));
        c.u_host_version = c.u_version;
    } else {
        // The skip is only right if EVERY device-side writer of the DoF vector bumped u_version. MISTARK_VERIFY_DOF_SKIP=1 (tests) does the
        // transfer anyway into a scratch buffer and compares it with what the caller's arrays hold: a writer that forgot shows up as an error
        // at the first skipped transfer instead of as callbacks silently reading old DoFs.
        static const bool verify = [] { const char* e = std::getenv("MISTARK_VERIFY_DOF_SKIP"); return e && e[0] == '1'; }();
        if (verify) {
            std::vector<double> tmp;
            for (auto& s : c.dof_sets) {
                if (s.n <= 0) continue;
                tmp.resize((size_t)s.n);
                MS_CHECK(hipMemcpyAsync(tmp.data(), c.u.p + s.offset, s.n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
                MS_CHECK(hipStreamSynchronize(c.stream));
                if (std::memcmp(tmp.data(), s.host, (size_t)s.n * sizeof(double)) != 0)
                    throw Error("mistark_dofs_to_host_arrays_if_changed: the transfer was skipped (version " + std::to_string(c.u_version) + ") but DoF set '" + s.label +
                                "' differs between the device and the caller's array: a writer of the DoF vector did not bump Context::u_version");
            }
            c.n_dof_skips_verified++;
        }
    }
    API_END(0)
}
int mistark_dofs_from_host_arrays(mistark_ctx* ctx)
{
    API_BEGIN
    ctx->c.touch();
    ctx->c.u_version++;
    Context& c = ctx->c;
    prepare(c);
    for (auto& s : c.dof_sets)
        if (s.n > 0) {
            (void)host_range_pinned(c, s.host, (size_t)s.n * sizeof(double));
            MS_CHECK(hipMemcpyAsync(c.u.p + s.offset, s.host, s.n * sizeof(double), hipMemcpyHostToDevice, c.stream));
        }
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}

int mistark_eval(mistark_ctx* ctx, int mode, double* E, double* grad_host)
{
    API_BEGIN
    if (mode < 0 || mode > 2) throw Error("bad eval mode");
    eval(ctx->c, mode, E, grad_host, nullptr, ctx->c.lazy_eval);
    API_END(0)
}
int mistark_get_element_hessians(mistark_ctx* ctx, int potential, double* values, int32_t* block_rows, int32_t* nb_out)
{
    API_BEGIN
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    if (!c.have_hessians) throw Error("no element Hessians");
    if (c.world > 1) throw Error("mistark_get_element_hessians: single-rank accessor (a sharded context holds the elements touching its rows)");
    Potential& P = c.pots[potential];
    if (values && c.lazy_active && P.lazy_capable) throw Error("element Hessians of '" + P.name + "' were evaluated on the lazy path (float blocks only); evaluate without lazy_eval");
    const int NB = P.NB, n = 3 * NB;
    if (nb_out) *nb_out = NB;
    if (values && P.n_elem > 0) {
        std::vector<double> tmp((size_t)P.n_elem * n * n);
        MS_CHECK(hipMemcpyAsync(tmp.data(), c.elemH.p + P.h_off, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
        for (int e = 0; e < P.n_elem; e++)
            for (int a = 0; a < NB; a++)
                for (int b = 0; b < NB; b++)
                    for (int i = 0; i < 3; i++)
                        for (int j = 0; j < 3; j++)
                            values[((size_t)e * n + 3 * a + i) * n + 3 * b + j] = tmp[((size_t)(a * NB + b) * P.n_elem + e) * 9 + i * 3 + j];
    }
    if (block_rows && P.n_elem > 0) {
        std::vector<int32_t> dev_conn;  // (tables of the device-side contact detector have no host copy)
        const int32_t* conn = P.conn_host.data();
        if (P.conn_ext) {
            dev_conn.resize((size_t)P.n_elem * P.conn_stride);
            MS_CHECK(hipMemcpyAsync(dev_conn.data(), P.conn_ext, dev_conn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
            MS_CHECK(hipStreamSynchronize(c.stream));
            conn = dev_conn.data();
        }
        for (int e = 0; e < P.n_elem; e++)
            for (int k = 0; k < NB; k++) block_rows[(size_t)e * NB + k] = P.args.dof_row_off[k] + conn[(size_t)e * P.conn_stride + P.args.dof_col[k]];
    }
    API_END(0)
}
int mistark_get_element_energies(mistark_ctx* ctx, int potential, double* values)
{
    API_BEGIN
    Context& c = ctx->c;
    if (potential < 0 || potential >= (int)c.pots.size()) throw Error("bad potential id");
    Potential& P = c.pots[potential];
    if (c.world > 1) throw Error("mistark_get_element_energies: single-rank accessor");
    if (P.n_elem > 0) {
        MS_CHECK(hipMemcpyAsync(values, c.elemE.p + P.e_off, (size_t)P.n_elem * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
    API_END(0)
}

int mistark_project(mistark_ctx* ctx, double eps, int mirroring, const uint8_t* active_blocks, int64_t* n_projected_now, int64_t* n_changed_now)
{
    API_BEGIN
    project(ctx->c, eps, mirroring, active_blocks, false, 0.0, nullptr, n_projected_now, n_changed_now);
    API_END(0)
}
int mistark_project_by_gradient(mistark_ctx* ctx, double eps, int mirroring, double threshold, int* all_active_out, int64_t* n_projected_now)
{
    API_BEGIN
    project(ctx->c, eps, mirroring, nullptr, true, threshold, all_active_out, n_projected_now, nullptr);
    API_END(0)
}
int mistark_assemble(mistark_ctx* ctx)
{
    API_BEGIN
    assemble(ctx->c);
    build_preconditioner(ctx->c);
    MS_CHECK(hipStreamSynchronize(ctx->c.stream));
    API_END(0)
}
int mistark_get_bsr(mistark_ctx* ctx, int64_t* n_block_rows, int64_t* nnzb, int64_t* row_ptr, int32_t* cols, float* vals)
{
    API_BEGIN
    Context& c = ctx->c;
    if (c.world > 1) throw Error("mistark_get_bsr: single-rank accessor (a sharded context holds its own block rows in local numbering)");
    ensure_pattern(c);
    // the engine keeps A = A_static + A_dynamic (contacts) as two block-CSR parts; this parity accessor merges them on the host
    struct Blk
    {
        uint64_t key;
        float v[9];
    };
    std::vector<Blk> blks;
    for (int part = 0; part < 2; part++) {
        const BsrPart& m = c.part[part];
        if (m.nnzb == 0) continue;
        std::vector<uint32_t> cw((size_t)m.nnzb), rw((size_t)m.nnzb);
        std::vector<float> tv;
        MS_CHECK(hipMemcpyAsync(cw.data(), m.colw.p, cw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipMemcpyAsync(rw.data(), m.slot_row.p, rw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
        if (vals) {
            if (!c.have_matrix) throw Error("matrix not assembled");
            tv.resize((size_t)m.ntiles * 576);
            MS_CHECK(hipMemcpyAsync(tv.data(), m.vals.p, tv.size() * sizeof(float), hipMemcpyDeviceToHost, c.stream));
        }
        std::vector<uint32_t> store;  // static part: position of CSR slot s in the chunk-aligned storage
        if (part == 0 && m.n_chunks_static > 0) {
            store.resize((size_t)m.nnzb);
            MS_CHECK(hipMemcpyAsync(store.data(), m.store_slot.p, store.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
        }
        MS_CHECK(hipStreamSynchronize(c.stream));
        for (int64_t s = 0; s < m.nnzb; s++) {
            Blk b{};
            // (solver numbering -> the caller's block rows: Context::perm_active)
            const uint64_t br = c.perm_active ? (uint64_t)c.iperm_h[(size_t)rw[s]] : (uint64_t)rw[s];
            const uint64_t bc = c.perm_active ? (uint64_t)c.iperm_h[(size_t)(cw[s] & 0x7fffffffu)] : (uint64_t)(cw[s] & 0x7fffffffu);
            b.key = br * (uint64_t)c.nbr + bc;
            if (vals) {
                const size_t pos = store.empty() ? (size_t)s : (size_t)store[s];
                const size_t base = (pos >> 6) * 576;
                const size_t lane = pos & 63;
                for (int k = 0; k < 9; k++) {
                    const size_t idx = k < 4 ? base + lane * 4 + k : (k < 8 ? base + 256 + lane * 4 + (k - 4) : base + 512 + lane);
                    b.v[k] = tv[idx];
                }
            }
            blks.push_back(b);
        }
    }
    std::stable_sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.key < b.key; });
    size_t n = 0;
    for (size_t i = 0; i < blks.size(); i++) {
        if (n > 0 && blks[n - 1].key == blks[i].key) {
            for (int k = 0; k < 9; k++) blks[n - 1].v[k] += blks[i].v[k];
        } else blks[n++] = blks[i];
    }
    blks.resize(n);
    if (n_block_rows) *n_block_rows = c.nbr;
    if (nnzb) *nnzb = (int64_t)n;
    if (row_ptr) {
        for (int64_t r = 0; r <= c.nbr; r++) row_ptr[r] = 0;
        for (const Blk& b : blks) row_ptr[b.key / (uint64_t)c.nbr + 1]++;
        for (int64_t r = 0; r < c.nbr; r++) row_ptr[r + 1] += row_ptr[r];
    }
    if (cols)
        for (size_t i = 0; i < n; i++) cols[i] = (int32_t)(blks[i].key % (uint64_t)c.nbr);
    if (vals)
        for (size_t i = 0; i < n; i++) std::memcpy(vals + 9 * i, blks[i].v, sizeof(float) * 9);
    API_END(0)
}
int mistark_spmv(mistark_ctx* ctx, const double* x_host, double* y_host)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!c.have_matrix) throw Error("matrix not assembled");
    MS_CHECK(hipMemcpyAsync(c.tmp_a.p, x_host, (size_t)c.ndofs * sizeof(double), hipMemcpyHostToDevice, c.stream));
    if (c.world > 1) {  // this rank's rows of y from its rows of A, gathered into the whole vector on every rank
        shard_to_local(c, c.tmp_a.p, c.p.p, true);
        spmv_device(c, c.p.p, c.q.p, nullptr, nullptr, true);
        shard_gather_global(c, c.q.p, c.tmp_b.p);
    } else if (c.perm_active) {  // the matrix lives in solver numbering
        rows_to_solver(c, c.tmp_a.p, c.p.p);
        spmv_device(c, c.p.p, c.q.p, nullptr, nullptr, true);
        rows_from_solver(c, c.q.p, c.tmp_b.p);
    } else {
        spmv_device(c, c.tmp_a.p, c.tmp_b.p, nullptr, nullptr, true);
    }
    MS_CHECK(hipMemcpyAsync(y_host, c.tmp_b.p, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
int mistark_apply_preconditioner(mistark_ctx* ctx, const double* x_host, double* z_host)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!c.have_matrix) throw Error("matrix not assembled");
    if (c.world > 1) throw Error("mistark_apply_preconditioner: single-rank accessor");
    build_preconditioner(c);
    // z = M^-1 x on the host from the device-built float inverse (parity probe only; the solver applies it on the device)
    std::vector<float> d((size_t)c.nbr * 9);
    MS_CHECK(hipMemcpyAsync(d.data(), c.dinv.p, d.size() * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    for (int64_t r = 0; r < c.nbr; r++) {
        const int64_t sr = c.perm_active ? (int64_t)c.perm_h[(size_t)r] : r;  // (the inverse blocks are stored by solver row)
        for (int i = 0; i < 3; i++) {
            double s = 0.0;
            for (int j = 0; j < 3; j++) s += (double)d[9 * sr + 3 * i + j] * x_host[3 * r + j];
            z_host[3 * r + i] = s;
        }
    }
    API_END(0)
}

int mistark_pcg(mistark_ctx* ctx, double abs_tol, double rel_tol, int max_iter, int stop_on_indefiniteness, double* du_host, mistark_pcg_info* info)
{
    API_BEGIN
    Context& c = ctx->c;
    vec_neg(c, c.tmp_a.p, c.grad.p, c.ndofs);
    pcg(c, c.tmp_a.p, abs_tol, rel_tol, max_iter, stop_on_indefiniteness, info);
    if (du_host) {
        MS_CHECK(hipMemcpyAsync(du_host, c.du.p, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
    API_END(0)
}
int mistark_pcg_rhs(mistark_ctx* ctx, const double* rhs_host, double abs_tol, double rel_tol, int max_iter, int stop_on_indefiniteness, double* x_host, mistark_pcg_info* info)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!rhs_host) throw Error("mistark_pcg_rhs: null right-hand side");
    if (c.dry) throw Error("mistark_pcg_rhs: a dry context (mistark_create_dry) has no device");
    if (!c.have_matrix) throw Error("mistark_pcg_rhs: matrix not assembled (mistark_assemble first)");
    c.tmp_a.ensure((size_t)c.ndofs);
    MS_CHECK(hipMemcpyAsync(c.tmp_a.p, rhs_host, (size_t)c.ndofs * sizeof(double), hipMemcpyHostToDevice, c.stream));
    pcg(c, c.tmp_a.p, abs_tol, rel_tol, max_iter, stop_on_indefiniteness, info);
    if (x_host) {
        MS_CHECK(hipMemcpyAsync(x_host, c.du.p, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
    API_END(0)
}

int mistark_direct_llt_rhs(mistark_ctx* ctx, const double* rhs_host, double* x_host, int* success)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!rhs_host || !success) throw Error("mistark_direct_llt_rhs: null argument");
    if (c.dry) throw Error("mistark_direct_llt_rhs: a dry context (mistark_create_dry) has no device");
    if (!c.have_matrix) throw Error("mistark_direct_llt_rhs: matrix not assembled (mistark_assemble first)");
    if (c.world > 1) throw Error("DirectLLT is a single-rank solver (a sharded context holds its own rows only); use the block-Jacobi PCG");
    c.tmp_a.ensure((size_t)c.ndofs);
    MS_CHECK(hipMemcpyAsync(c.tmp_a.p, rhs_host, (size_t)c.ndofs * sizeof(double), hipMemcpyHostToDevice, c.stream));
    *success = direct_llt(c, c.tmp_a.p, c.du.p) ? 1 : 0;
    if (x_host) {
        MS_CHECK(hipMemcpyAsync(x_host, c.du.p, (size_t)c.ndofs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
    API_END(0)
}

void mistark_newton_default_settings(mistark_newton_settings* s)
{
    if (!s) return;
    // symx::NewtonSettings defaults (solver_utils.h:173-259) with STARK's overrides (stark/src/core/Settings.cpp:43-50)
    s->max_iterations = 2147483647;
    s->min_iterations = 0;
    s->residual_tolerance_abs = 1e-6;
    s->residual_tolerance_rel = 0.0;
    s->step_tolerance = 1e-3;
    s->max_iterations_as_success = 0;
    s->step_cap = 1e300 * 1e300;  // +inf
    s->enable_armijo_backtracking = 1;
    s->line_search_armijo_beta = 1e-4;
    s->max_backtracking_armijo_iterations = 20;
    s->max_backtracking_invalid_state_iterations = 8;
    s->projection_mode = MISTARK_PROJ_PROGRESSIVE;
    s->projection_eps = 1e-10;
    s->project_to_pd_use_mirroring = 0;
    s->project_on_demand_countdown = 4;
    s->ppn_tightening_factor = 0.5;
    s->ppn_release_factor = 2.0;
    s->cg_max_iterations = 10000;
    s->cg_abs_tolerance = 1e-12;
    s->cg_rel_tolerance = 1e-4;
    s->cg_stop_on_indefiniteness = 1;
    s->bailout_residual = 1e-10;
    s->linear_solver = MISTARK_SOLVER_BDPCG;
}

int mistark_newton_iteration_log(mistark_ctx* ctx, mistark_newton_iteration* out, int32_t cap, int32_t* n)
{
    API_BEGIN
    const std::vector<mistark_newton_iteration>& log = ctx->c.newton_log;
    if (n) *n = (int32_t)log.size();
    if (out && cap > 0) std::memcpy(out, log.data(), sizeof(mistark_newton_iteration) * std::min<size_t>((size_t)cap, log.size()));
    API_END(0)
}
int mistark_newton_solve(mistark_ctx* ctx, const mistark_newton_settings* settings, const mistark_newton_callbacks* callbacks, mistark_newton_stats* stats)
{
    API_BEGIN
    mistark_newton_settings s;
    if (settings) s = *settings;
    else mistark_newton_default_settings(&s);
    mistark_newton_stats st{};
    const int r = newton_solve(ctx->c, s, callbacks, st);
    if (stats) *stats = st;
    return r;
    API_END(0)
}

// ---- multi-GPU -------------------------------------------------------------------------------------------------------------------
struct mistark_local_group
{
    std::shared_ptr<LocalGroup> g;
};
int mistark_shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* end)
{
    if (world < 1 || rank < 0 || rank >= world || n < 0) return -1;
    long long b, e;
    shard_range(n, rank, world, b, e);
    if (begin) *begin = b;
    if (end) *end = e;
    return 0;
}
int mistark_partition_rows(int64_t n_block_rows, int world, int n_tables, const int32_t* const* rows, const int64_t* n_elem, const int32_t* nb, const uint8_t* hub, int32_t* owner_out)
{
    if (n_block_rows <= 0 || world < 1 || n_tables < 0 || !owner_out) return -1;
    if (n_tables > 0 && (!rows || !n_elem || !nb)) return -1;
    for (int t = 0; t < n_tables; t++)
        if (n_elem[t] < 0 || nb[t] <= 0 || (n_elem[t] > 0 && !rows[t])) return -1;
    try {
        std::vector<ElemTable> tables;
        for (int t = 0; t < n_tables; t++) tables.push_back(ElemTable{rows[t], n_elem[t], nb[t]});
        std::vector<int32_t> owner;
        graph_partition_rows(n_block_rows, world, tables, hub, owner);
        std::memcpy(owner_out, owner.data(), (size_t)n_block_rows * sizeof(int32_t));
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}
int mistark_dist_unique_id(char out[128])
{
    try {
        rccl_unique_id(out);
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}
static void set_dist(Context& c, int rank, int world, std::unique_ptr<Collective> coll)
{
    if (world < 1 || rank < 0 || rank >= world) throw Error("bad rank / world size");
    c.rank = rank;
    c.world = world;
    c.coll = std::move(coll);
    if (c.coll && c.coll->shared_stream()) {  // in-process group: all ranks run on the group's stream
        MS_CHECK(hipStreamSynchronize(c.stream));
        if (c.owns_stream && c.stream) (void)hipStreamDestroy(c.stream);
        c.stream = c.coll->shared_stream();
        c.owns_stream = false;
    }
    c.sh.sig.clear();
    c.layout_dirty = true;
    c.part[0].dirty = c.part[1].dirty = true;
}
int mistark_dist_init_rccl(mistark_ctx* ctx, int rank, int world, const char unique_id[128])
{
    API_BEGIN
    MS_CHECK(hipSetDevice(ctx->c.device));
    set_dist(ctx->c, rank, world, world > 1 ? make_rccl_collective(rank, world, unique_id) : nullptr);
    API_END(0)
}
int mistark_dist_rccl_selftest(mistark_ctx* ctx, double* inout, int64_t n)
{
    // one-rank communicator through the same dlopen'ed entry points the N-rank path uses: the all-gather of one rank returns its input
    API_BEGIN
    Context& c = ctx->c;
    MS_CHECK(hipSetDevice(c.device));
    char id[128];
    rccl_unique_id(id);
    std::unique_ptr<Collective> coll = make_rccl_collective(0, 1, id);
    DevBuf<double> d, o;
    d.ensure((size_t)n);
    o.ensure((size_t)n);
    MS_CHECK(hipMemcpyAsync(d.p, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipMemsetAsync(o.p, 0, (size_t)n * sizeof(double), c.stream));
    coll->allgather_f64(d.p, o.p, (size_t)n, c.stream);
    MS_CHECK(hipMemcpyAsync(inout, o.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    API_END(0)
}
int mistark_dist_move(mistark_ctx* ctx, mistark_ctx* from)
{
    // a scene that registers again (objects added mid-run) builds a new context: the communicator moves over (an RCCL unique id is
    // single-use, a second ncclCommInitRank on it would hang)
    API_BEGIN
    if (!from) throw Error("mistark_dist_move: null source");
    Context& o = from->c;
    if (o.world > 1 && !o.coll) throw Error("mistark_dist_move: the source has no communicator");
    MS_CHECK(hipStreamSynchronize(o.stream));
    set_dist(ctx->c, o.rank, o.world, std::move(o.coll));
    ctx->c.sh.user_owner = o.sh.user_owner;
    ctx->c.sh.coords = o.sh.coords;
    o.world = 1;
    o.rank = 0;
    API_END(0)
}
int mistark_dist_set_row_owner(mistark_ctx* ctx, const int32_t* owner, int64_t n_block_rows)
{
    API_BEGIN
    Context& c = ctx->c;
    if (owner && n_block_rows > 0) c.sh.user_owner.assign(owner, owner + n_block_rows);
    else c.sh.user_owner.clear();
    c.sh.version++;
    c.layout_dirty = true;
    API_END(0)
}
int64_t mistark_dof_set_first_row(mistark_ctx* ctx, int set)
{
    if (!ctx || set < 0 || set >= (int)ctx->c.dof_sets.size()) return -1;
    int64_t off = 0;
    for (int s = 0; s < set; s++) off += ctx->c.dof_sets[(size_t)s].n;
    return off / 3;
}
int mistark_dist_set_row_coords(mistark_ctx* ctx, const double* xyz, int64_t n_block_rows)
{
    API_BEGIN
    Context& c = ctx->c;
    if (xyz && n_block_rows > 0) c.sh.coords.assign(xyz, xyz + 3 * n_block_rows);
    else c.sh.coords.clear();
    c.sh.version++;
    c.layout_dirty = true;
    API_END(0)
}
int mistark_dist_add_shared_rows(mistark_ctx* ctx, const int32_t* rows, int64_t n)
{
    API_BEGIN
    Context& c = ctx->c;
    if (rows && n > 0) c.sh.shared_rows.insert(c.sh.shared_rows.end(), rows, rows + n);
    c.sh.version++;
    c.layout_dirty = true;
    API_END(0)
}
int mistark_dist_info(mistark_ctx* ctx, int64_t* out, int n)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!out && n > 0) throw Error("mistark_dist_info: null output");
    prepare(c);
    int64_t n_all = 0, n_here = 0;  // static potentials: elements registered / elements this rank evaluates (interface elements on every side)
    for (const Potential& P : c.pots)
        if (P.part == 0) {
            n_all += P.n_elem;
            n_here += c.world > 1 ? P.n_key : P.n_elem;
        }
    const int64_t v[15] = {c.world > 1 ? c.sh.n_own : c.nbr, c.world > 1 ? c.sh.n_ghost : 0, c.world > 1 ? c.sh.n_send : 0, (int64_t)c.n_elem_total, c.part[0].nnzb, c.part[1].nnzb,
                           c.n_fused_solves, c.n_unfused_solves, c.world, c.rank, c.coll ? c.coll->transport_id() : 0, c.coll ? c.coll->transport_ranks() : 1,
                           contact_sharded_searches(c), n_all, n_here};
    for (int i = 0; i < n && i < 15; i++) out[i] = v[i];
    API_END(0)
}
int mistark_dist_get_row_owner(mistark_ctx* ctx, int32_t* owner)
{
    API_BEGIN
    Context& c = ctx->c;
    if (!owner) throw Error("mistark_dist_get_row_owner: null output");
    prepare(c);
    for (int64_t r = 0; r < c.nbr; r++) owner[r] = c.world > 1 ? c.sh.owner[(size_t)r] : 0;
    API_END(0)
}
// ---- IPC windows: one process per rank, exchanges by stores into the peers' windows (dist.hip) ----
struct mistark_ipc_comm
{
    std::shared_ptr<IpcComm> m;
    std::string last_error;
};
mistark_ipc_comm* mistark_ipc_comm_create(int device, int rank, int world, int64_t window_bytes, char handle_out[64])
{
    if (!handle_out) return nullptr;
    try {
        auto* h = new mistark_ipc_comm();
        try {
            h->m = ipc_comm_create(device, rank, world, (size_t)std::max<int64_t>(window_bytes, 0), handle_out);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "mistark_ipc_comm_create: %s\n", e.what());
            delete h;
            return nullptr;
        }
        return h;
    } catch (...) {
        return nullptr;
    }
}
int mistark_ipc_comm_connect(mistark_ipc_comm* comm, const char* handles, int64_t n_bytes)
{
    if (!comm || !comm->m) return -1;
    try {
        if (!handles || n_bytes != (int64_t)ipc_comm_world(*comm->m) * 64) throw Error("mistark_ipc_comm_connect: world x 64 bytes of handles, in rank order");
        ipc_comm_connect(*comm->m, handles);
    } catch (const std::exception& e) {
        comm->last_error = e.what();
        return -1;
    }
    return 0;
}
const char* mistark_ipc_comm_last_error(mistark_ipc_comm* comm) { return comm ? comm->last_error.c_str() : "null communicator"; }
void mistark_ipc_comm_destroy(mistark_ipc_comm* comm) { delete comm; }
int mistark_dist_init_ipc(mistark_ctx* ctx, mistark_ipc_comm* comm)
{
    API_BEGIN
    if (!comm || !comm->m) throw Error("null communicator");
    if (ipc_comm_device(*comm->m) != ctx->c.device) throw Error("mistark_dist_init_ipc: the communicator's window lives on another device than the context");
    set_dist(ctx->c, ipc_comm_rank(*comm->m), ipc_comm_world(*comm->m), ipc_comm_world(*comm->m) > 1 ? make_ipc_collective(comm->m) : nullptr);
    API_END(0)
}
// `iters` all-gathers of n doubles with values every rank can predict, checked on the device's results; avg_us[0]: the average wall time of
// one exchange followed by a stream synchronisation, avg_us[1]: of one exchange in a train of `iters` enqueued back to back (microseconds).
// Every rank must call it with the same arguments.
int mistark_ipc_comm_selftest(mistark_ipc_comm* comm, int64_t n, int iters, double* avg_us)
{
    if (!comm || !comm->m || n <= 0 || iters <= 0) return -1;
    try {
        IpcComm& m = *comm->m;
        const int W = ipc_comm_world(m), me = ipc_comm_rank(m);
        MS_CHECK(hipSetDevice(ipc_comm_device(m)));
        std::unique_ptr<Collective> coll = make_ipc_collective(comm->m);
        hipStream_t s;
        MS_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        DevBuf<double> send, recv;
        send.ensure((size_t)n);
        recv.ensure((size_t)n * (size_t)W);
        std::vector<double> h((size_t)n), out((size_t)n * (size_t)W);
        double total = 0.0;
        for (int it = 0; it < iters; it++) {
            for (int64_t i = 0; i < n; i++) h[(size_t)i] = 1e6 * me + 1e3 * it + (double)i + 0.25;
            MS_CHECK(hipMemcpyAsync(send.p, h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
            MS_CHECK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            coll->allgather_f64(send.p, recv.p, (size_t)n, s);
            MS_CHECK(hipStreamSynchronize(s));
            total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            coll->check();
            MS_CHECK(hipMemcpy(out.data(), recv.p, out.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int r = 0; r < W; r++)
                for (int64_t i = 0; i < n; i++)
                    if (out[(size_t)r * (size_t)n + (size_t)i] != 1e6 * r + 1e3 * it + (double)i + 0.25)
                        throw Error("IPC self-test: rank " + std::to_string(me) + " received a wrong value from rank " + std::to_string(r) + " (exchange " + std::to_string(it) + ", entry " +
                                    std::to_string(i) + ")");
        }
        // the same exchanges back to back, one synchronisation at the end: what an exchange costs a stream that keeps running (a rank's push
        // is enqueued behind its previous wait: the figure holds the one-way latency, two kernel boundaries and the ranks' skew)
        MS_CHECK(hipStreamSynchronize(s));
        const auto t1 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; it++) coll->allgather_f64(send.p, recv.p, (size_t)n, s);
        MS_CHECK(hipStreamSynchronize(s));
        const double piped = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        coll->check();
        (void)hipStreamDestroy(s);
        if (avg_us) {
            avg_us[0] = 1e6 * total / iters;
            avg_us[1] = 1e6 * piped / iters;
        }
    } catch (const std::exception& e) {
        comm->last_error = e.what();
        return -1;
    }
    return 0;
}
int mistark_ipc_comm_preflight(mistark_ipc_comm* comm, int iters, double timeout_s, double* half_rtt_us)
{
    if (!comm || !comm->m || !half_rtt_us) return -1;
    try {
        return ipc_comm_preflight(*comm->m, iters, timeout_s, half_rtt_us);
    } catch (const std::exception& e) {
        comm->last_error = e.what();
        return -1;
    }
}
int mistark_rccl_allreduce_bench(int device, int rank, int world, const char unique_id[128], int64_t n_big, int reps, double* out, char* err, int err_len)
{
    if (!unique_id || !out) return -1;
    try {
        rccl_allreduce_bench(device, rank, world, unique_id, (size_t)std::max<int64_t>(n_big, 0), reps, out);
        return 0;
    } catch (const std::exception& e) {
        if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", e.what());
        return -1;
    }
}
// Solo durations (microseconds) of the fused PCG iteration's two kernels on THIS rank's shard: out[0] = S (SpMV with halo polls), out[1] = V
// (vector kernel; its workgroup 0 reduces and pushes the rank's sums), replayed from the last converged solve. Not a collective: the caller lets the
// ranks take turns (bench.py: torch.distributed barriers between them), so that on a box where all ranks share ONE GPU the other ranks' queues
// are empty while one measures; no rank may start a solve in between.
int mistark_dist_fused_bench(mistark_ctx* ctx, int n_launches, double* out)
{
    API_BEGIN
    Context& c = ctx->c;
    if (c.world < 2 || !c.coll || !c.coll->ipc()) throw Error("mistark_dist_fused_bench: ranks that exchange through windows only");
    if (!out || n_launches <= 0) throw Error("mistark_dist_fused_bench: bad arguments");
    fused_pcg_replay(c, n_launches, &out[0], &out[1]);
    API_END(0)
}
mistark_local_group* mistark_local_group_create(int world)
{
    if (world < 1) return nullptr;
    auto* g = new mistark_local_group();
    g->g = std::make_shared<LocalGroup>(world);
    return g;
}
void mistark_local_group_destroy(mistark_local_group* g) { delete g; }
int mistark_dist_init_local(mistark_ctx* ctx, mistark_local_group* group, int rank)
{
    API_BEGIN
    if (!group) throw Error("null group");
    set_dist(ctx->c, rank, group->g->world, group->g->world > 1 ? make_local_collective(group->g, rank, ctx->c.device) : nullptr);
    API_END(0)
}

int mistark_set_option(mistark_ctx* ctx, const char* name, int value)
{
    API_BEGIN
    const std::string n = name ? name : "";
    if (n == "force_generic") {
        ctx->c.force_generic = value != 0;
        ctx->c.layout_dirty = true;  // (Potential::lazy_capable depends on it)
    }
    else if (n == "generic_contact") ctx->c.generic_contact = value != 0;
    else if (n == "no_eval_prelaunch") ctx->c.no_eval_prelaunch = value != 0;
    else if (n == "no_multi_eval_p") ctx->c.no_multi_eval_p = value != 0;
    else if (n == "no_multi_eval_pgh") ctx->c.no_multi_eval_pgh = value != 0;
    else if (n == "late_eager_assembly") ctx->c.late_eager_assembly = value != 0;
    else if (n == "llt_multifrontal") ctx->c.llt_multifrontal = value;
    else if (n == "llt_no_coords") { ctx->c.llt_no_coords = value != 0; ctx->c.llt_mf_pattern_version = 0; }
    else if (n == "pcg_holdback") ctx->c.pcg_holdback = value != 0;
    else if (n == "sweep_axis_by_extent") ctx->c.sweep_axis_by_extent = value != 0;
    else if (n == "pin_host_arrays") ctx->c.pin_host_arrays = value != 0;
    else if (n == "generic_inertia") ctx->c.generic_inertia = value != 0;
    else if (n == "contact_closed_min_lanes") ctx->c.contact_closed_min_lanes = value;
    else if (n == "atomic_assembly") ctx->c.atomic_assembly = value != 0;
    else if (n == "spmv_grid_cap") ctx->c.spmv_grid_cap = value;
    else if (n == "spmv_nt") ctx->c.spmv_nt = value;
    else if (n == "custom_rtc") ctx->c.custom_rtc = value;
    else if (n == "hf_layout") {  // (the pool is rewritten by the next evaluation; the gather's descriptors follow the layout)
        ctx->c.hf_layout = value;
        ctx->c.part[0].desc_lazy = ctx->c.part[1].desc_lazy = -1;
        ctx->c.matrix_current = false;
        ctx->c.static_assembled = false;
    }
    else if (n == "custom_timing") ctx->c.custom_timing = value;
    else if (n == "pcg_batch") ctx->c.pcg_batch = value;
    else if (n == "lazy_hessians") ctx->c.lazy_allowed = value != 0;  // newton_solve: float upper-triangle pool for the closed-form tets
    else if (n == "kernel_dbg") { ctx->c.kernel_dbg = value; ctx->c.layout_dirty = true; }  // measurement only
    else if (n == "lazy_eval") ctx->c.lazy_eval = value != 0;
    else if (n == "no_pattern_overlap") ctx->c.no_pattern_overlap = value != 0;
    else if (n == "no_eval_overlap") ctx->c.no_eval_overlap = value != 0;
    else if (n == "contact_speculation") ctx->c.contact_speculation = value != 0;
    else if (n == "no_eager_assembly") ctx->c.no_eager_assembly = value != 0;
    else if (n == "no_sym_gather") ctx->c.no_sym_gather = value != 0;
    else if (n == "no_split_gather") ctx->c.no_split_gather = value != 0;
    else if (n == "no_bounded_pattern") { ctx->c.no_bounded_pattern = value != 0; ctx->c.part[1].dirty = true; }
    else if (n == "fuse_dir") ctx->c.no_fuse_dir = value == 0;
    else if (n == "no_grad_gather") { ctx->c.no_grad_gather = value != 0; ctx->c.layout_dirty = true; }         // staged mistark_eval calls take the lazy path too (tests)

    else if (n == "spmv_variant") ctx->c.spmv_variant = value;
    else if (n == "proj_variant") ctx->c.proj_variant = value;
    else if (n == "no_contact_cache") ctx->c.no_contact_cache = value != 0;
    else if (n == "no_sharded_search") ctx->c.no_sharded_search = value != 0;
    else if (n == "seg_sort") ctx->c.seg_sort = value != 0;
    else if (n == "proj_speculation") ctx->c.proj_speculation = value != 0;
    else if (n == "no_row_order") { ctx->c.no_row_order = value != 0; ctx->c.perm_sig.clear(); ctx->c.layout_dirty = true; }  // one GPU: the caller's row numbering in the solver, too
    else if (n == "row_order") { ctx->c.row_order_mode = value; ctx->c.perm_sig.clear(); ctx->c.layout_dirty = true; }
    else if (n == "atomic_projection") ctx->c.atomic_projection = value != 0;
    else if (n == "no_dyn_pool") { ctx->c.no_dyn_pool = value != 0; ctx->c.layout_dirty = true; }  // gradient of device-resident tables by atomics (arrival order) instead of the sorted gather
    else if (n == "no_halo_subset") { ctx->c.no_halo_subset = value != 0; ctx->c.cg_mask_pattern = 0; }
    else if (n == "cg_variant") ctx->c.cg_variant = value;
    else if (n == "no_fused_pcg") ctx->c.no_fused_pcg = value != 0;  // sharded runs over windows: the five-launch iteration with two all-gathers instead of the fused one
    else if (n == "spmv_chunk_tiles") {
        if (value < 0 || value > 64) throw Error("spmv_chunk_tiles: 0 (automatic) .. 64");
        ctx->c.spmv_chunk_tiles = value;
        ctx->c.part[0].dirty = true;  // the storage layout depends on it
    }
    else throw Error("unknown option '" + n + "'");
    API_END(0)
}

int mistark_sync(mistark_ctx* ctx)
{
    API_BEGIN
    MS_CHECK(hipStreamSynchronize(ctx->c.stream));
    API_END(0)
}
int mistark_spmv_bench(mistark_ctx* ctx, int n_launches, double* avg_us)
{
    API_BEGIN
    const double us = spmv_bench(ctx->c, n_launches);
    if (avg_us) *avg_us = us;
    API_END(0)
}

int mistark_spmv_event_overhead(mistark_ctx* ctx, double* avg_ms)
{
    API_BEGIN
    Context& c = ctx->c;
    if (avg_ms) *avg_ms = c.spmv_n > 0 ? c.spmv_empty_ms_sum / (double)c.spmv_n : 0.0;
    API_END(0)
}
int mistark_spmv_device_clock(mistark_ctx* ctx, double* avg_ms, int64_t* n)
{
    API_BEGIN
    Context& c = ctx->c;
    int khz = 0;
    MS_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c.device));
    if (khz <= 0) throw Error("spmv_device_clock: the device reports no wall clock rate");
    if (avg_ms) *avg_ms = c.spmv_clk_n > 0 ? c.spmv_clk_ticks / (double)c.spmv_clk_n / (double)khz : 0.0;
    if (n) *n = c.spmv_clk_n;
    API_END(0)
}
int mistark_spmv_timing(mistark_ctx* ctx, int reset, double* avg_ms, int64_t* n, double* bytes_per_launch)
{
    API_BEGIN
    Context& c = ctx->c;
    if (avg_ms) *avg_ms = c.spmv_n > 0 ? c.spmv_ms_sum / (double)c.spmv_n : 0.0;
    if (n) *n = c.spmv_n;
    // algorithmic bytes of one SpMV: nnzb*(9*4+4) + (nbr+1)*8 + 2*(3*nbr)*8
    if (bytes_per_launch)
        *bytes_per_launch = (double)(c.part[0].nnzb + c.part[1].nnzb) * 40.0 + ((double)c.nbr + 1.0) * 8.0 + 48.0 * (double)c.nbr + 56.0 * (double)c.part[1].n_rows;
    if (reset) {
        c.spmv_ms_sum = 0.0;
        c.spmv_empty_ms_sum = 0.0;
        c.spmv_n = 0;
        c.spmv_clk_ticks = 0.0;
        c.spmv_clk_n = 0;
        c.time_spmv = reset > 0;
    }
    API_END(0)
}

}  // extern "C"
