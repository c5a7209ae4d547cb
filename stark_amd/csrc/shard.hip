// shard.hip — the sharded problem (world > 1): row partition, local numbering, element lists and the exchanges.
//
// * Block rows are partitioned over the ranks by an owner map: given by the caller (mistark_dist_set_row_owner, e.g. slabs from the
//   host mirror's positions) or computed here from the connectivity graph of the potentials (breadth-first levels from a
//   pseudo-peripheral row, cut by element incidences: contiguous, slab-like parts).
// * A rank evaluates every element that touches one of ITS rows. Elements on an interface are evaluated by both sides, bit for bit
//   the same numbers, so gradient rows and matrix rows of a rank are complete without any gradient / Hessian traffic, and its matrix
//   rows are summed in the same order as on one GPU. An element's ENERGY counts on the rank that owns the row of its first DoF block.
// * Local numbering of a rank: its rows in ascending global order, then the ghosts (rows of other ranks its elements touch, plus the
//   rows potentials with changing connectivity may reference: contact surfaces, rigid bodies) grouped by owner. The matrix a rank holds
//   has n_own block rows and n_own + n_ghost block columns; the PCG vectors live in this numbering.
// * Exchanges, all through Collective::allgather_f64: boundary values (owners -> ghosts) of the search direction in every CG iteration
//   and of the gradient after an evaluation, the owned parts of the solution, and a handful of scalars (energy, dot products, counts),
//   which every rank reduces in rank order (identical bits everywhere).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <queue>

#include "dist.hpp"
#include "engine.hpp"

namespace mistark {
namespace {
constexpr int TB = 256;
inline int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + TB - 1) / TB, 1 << 20)); }

__global__ __launch_bounds__(TB) void k_pack_rows(const double* __restrict__ v, const int32_t* __restrict__ rows, const int32_t* __restrict__ grow, int64_t n, double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= 3 * n) return;
    const int64_t i = t / 3;
    int64_t r = rows[i];
    if (grow) r = grow[r];  // v is in global numbering
    out[t] = v[3 * r + (t - 3 * i)];
}
__global__ __launch_bounds__(TB) void k_unpack_ghosts(const double* __restrict__ recv, const int32_t* __restrict__ ghost_src, int64_t n_ghost, int64_t n_own,
                                                      const int32_t* __restrict__ grow, double* __restrict__ v)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= 3 * n_ghost) return;
    const int64_t i = t / 3;
    const int c = (int)(t - 3 * i);
    const int64_t dst = grow ? (int64_t)grow[n_own + i] : n_own + i;
    v[3 * dst + c] = recv[3 * (int64_t)ghost_src[i] + c];
}
__global__ __launch_bounds__(TB) void k_to_local(const double* __restrict__ vg, const int32_t* __restrict__ grow, int64_t n, double* __restrict__ vl)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= 3 * n) return;
    const int64_t i = t / 3;
    vl[t] = vg[3 * (int64_t)grow[i] + (t - 3 * i)];
}
__global__ __launch_bounds__(TB) void k_copy_pad(const double* __restrict__ v, int64_t n, int64_t n_pad, double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n_pad) return;
    out[t] = t < n ? v[t] : 0.0;
}
__global__ __launch_bounds__(TB) void k_scatter_all(const double* __restrict__ recv, const int32_t* __restrict__ grow_all, int64_t n, double* __restrict__ vg)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= 3 * n) return;
    const int64_t i = t / 3;
    const int32_t g = grow_all[i];
    if (g >= 0) vg[3 * (int64_t)g + (t - 3 * i)] = recv[t];
}

// global block rows of element e of potential P (host connectivity)
inline int64_t row_of(const Potential& P, int e, int k) { return (int64_t)P.args.dof_row_off[k] + P.conn_host[(size_t)e * P.conn_stride + P.args.dof_col[k]]; }
inline bool has_host_conn(const Potential& P) { return !P.conn_ext && P.n_elem > 0 && !P.conn_host.empty(); }
// What the partition, the ghost sets and the element lists are made from: potentials with FIXED host connectivity. A potential whose table is
// refilled inside the Newton loop (part 1: contact tables handed over by a caller, e.g. through the SymX shim) is treated like the device-resident
// tables: every rank evaluates all of its rows, the rows it references must be shared rows (mistark_dist_add_shared_rows) — its refills
// neither move the owner map nor rebuild the static pattern.
inline bool shards_by_list(const Potential& P) { return has_host_conn(P) && P.part == 0; }

}  // namespace

// Owner map from the connectivity graph: breadth-first order from a pseudo-peripheral row (two sweeps), components one after the other,
// cut into `world` consecutive pieces of equal weight (element incidences). Hub rows (rows of small DoF sets: rigid bodies, which would
// put the whole mesh within two hops) are left out of the graph and given to the last rank. Pure host code (mistark_partition_rows).
// (the traversal itself: rows in breadth-first order, hubs left out; weight[r] = 1 + element incidences of row r)
void graph_bfs_order(int64_t nbr, const std::vector<ElemTable>& tables, const uint8_t* hub, std::vector<int64_t>& order, std::vector<int64_t>& weight)
{
    auto is_hub = [&](int64_t r) { return hub && hub[(size_t)r]; };
    std::vector<int64_t> tab_base;
    int64_t n_el = 0;
    for (const ElemTable& T : tables) {
        tab_base.push_back(n_el);
        n_el += T.n_elem;
    }
    std::vector<int64_t> start((size_t)nbr + 1, 0);
    for (const ElemTable& T : tables)
        for (int64_t i = 0; i < T.n_elem * T.nb; i++) {
            if (T.rows[i] < 0 || T.rows[i] >= nbr) throw Error("partition: block row out of range");
            start[(size_t)T.rows[i] + 1]++;
        }
    weight.assign((size_t)nbr, 0);
    for (int64_t r = 0; r < nbr; r++) {
        weight[(size_t)r] = 1 + start[(size_t)r + 1];
        start[(size_t)r + 1] += start[(size_t)r];
    }
    std::vector<int64_t> inc((size_t)start[(size_t)nbr]);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (size_t ti = 0; ti < tables.size(); ti++)
            for (int64_t e = 0; e < tables[ti].n_elem; e++)
                for (int k = 0; k < tables[ti].nb; k++) inc[(size_t)fill[(size_t)tables[ti].rows[e * tables[ti].nb + k]]++] = tab_base[ti] + e;
    }
    std::vector<int32_t> level((size_t)nbr, -1);
    order.clear();
    order.reserve((size_t)nbr);
    // breadth-first sweep from `root` over unvisited (level < 0) non-hub rows; appends to out; returns the last row reached
    auto bfs = [&](int64_t root, std::vector<int64_t>& out) {
        const size_t first = out.size();
        out.push_back(root);
        level[(size_t)root] = 0;
        for (size_t h = first; h < out.size(); h++) {
            const int64_t u = out[h];
            for (int64_t j = start[(size_t)u]; j < start[(size_t)u + 1]; j++) {
                const int64_t id = inc[(size_t)j];
                const size_t ti = (size_t)(std::upper_bound(tab_base.begin(), tab_base.end(), id) - tab_base.begin()) - 1;
                const ElemTable& T = tables[ti];
                const int32_t* rows = T.rows + (id - tab_base[ti]) * T.nb;
                for (int k = 0; k < T.nb; k++) {
                    const int64_t v = rows[k];
                    if (level[(size_t)v] < 0 && !is_hub(v)) {
                        level[(size_t)v] = level[(size_t)u] + 1;
                        out.push_back(v);
                    }
                }
            }
        }
        return out.back();
    };
    for (int64_t r0 = 0; r0 < nbr; r0++) {
        if (level[(size_t)r0] >= 0 || is_hub(r0)) continue;
        // pseudo-peripheral start: the far end of a sweep from r0, and the far end of a sweep from there
        std::vector<int64_t> tmp;
        int64_t far = bfs(r0, tmp);
        for (int64_t v : tmp) level[(size_t)v] = -1;
        tmp.clear();
        far = bfs(far, tmp);
        for (int64_t v : tmp) level[(size_t)v] = -1;
        bfs(far, order);
    }
}
void graph_partition_rows(int64_t nbr, int W, const std::vector<ElemTable>& tables, const uint8_t* hub, std::vector<int32_t>& owner)
{
    std::vector<int64_t> order, weight;
    graph_bfs_order(nbr, tables, hub, order, weight);
    int64_t total = 0;
    for (int64_t v : order) total += weight[(size_t)v];
    owner.assign((size_t)nbr, (int32_t)(W - 1));  // hubs (and nothing else) keep the last rank
    int64_t acc = 0;
    for (int64_t v : order) {
        owner[(size_t)v] = (int32_t)std::min<int64_t>(W - 1, total > 0 ? acc * W / total : 0);
        acc += weight[(size_t)v];
    }
}

// Owner map from positions: recursive coordinate bisection. The rows with a position are split along the longest axis of their bounding box
// into two parts whose weights (element incidences) are in the ratio of the rank counts they go to, recursively; rows without one (NaN:
// rigid bodies) keep the last rank. Compact, box-like parts: fewer interface rows than the level sets of the graph partition.
void rcb_partition_rows(int64_t nbr, int W, const double* xyz, const std::vector<int64_t>& weight, std::vector<int32_t>& owner)
{
    owner.assign((size_t)nbr, (int32_t)(W - 1));
    std::vector<int64_t> idx;
    for (int64_t r = 0; r < nbr; r++)
        if (xyz[3 * r] == xyz[3 * r] && xyz[3 * r + 1] == xyz[3 * r + 1] && xyz[3 * r + 2] == xyz[3 * r + 2]) idx.push_back(r);
    struct Job
    {
        size_t b, e;
        int r0, nr;
    };
    std::vector<Job> jobs{{0, idx.size(), 0, W}};
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        if (j.nr == 1 || j.e - j.b <= 1) {
            for (size_t i = j.b; i < j.e; i++) owner[(size_t)idx[i]] = (int32_t)j.r0;
            continue;
        }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = j.b; i < j.e; i++)
            for (int d = 0; d < 3; d++) {
                lo[d] = std::min(lo[d], xyz[3 * idx[i] + d]);
                hi[d] = std::max(hi[d], xyz[3 * idx[i] + d]);
            }
        int ax = 0;
        for (int d = 1; d < 3; d++)
            if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
        std::sort(idx.begin() + (long)j.b, idx.begin() + (long)j.e, [&](int64_t a, int64_t b) {
            const double xa = xyz[3 * a + ax], xb = xyz[3 * b + ax];
            return xa < xb || (xa == xb && a < b);
        });
        int64_t total = 0;
        for (size_t i = j.b; i < j.e; i++) total += weight[(size_t)idx[i]];
        const int n1 = j.nr / 2;
        const int64_t target = total * n1 / j.nr;
        int64_t acc = 0;
        size_t cut = j.b;
        while (cut < j.e && acc < target) acc += weight[(size_t)idx[cut++]];
        cut = std::min(std::max(cut, j.b + 1), j.e - 1);
        jobs.push_back({j.b, cut, j.r0, n1});
        jobs.push_back({cut, j.e, j.r0 + n1, j.nr - n1});
    }
}

namespace {
void graph_partition(Context& c, std::vector<int32_t>& owner)
{
    const int64_t nbr = c.nbr;
    std::vector<uint8_t> hub((size_t)nbr, 0);
    for (auto& s : c.dof_sets) {
        const int64_t rows = s.n / 3;
        if (rows > 0 && rows <= HOT_SET_ROWS)
            for (int64_t r = 0; r < rows; r++) hub[(size_t)(s.offset / 3 + r)] = 1;
    }
    std::vector<std::vector<int32_t>> rows;
    std::vector<ElemTable> tables;
    for (auto& P : c.pots)
        if (P.part == 0 && has_host_conn(P)) {
            rows.emplace_back((size_t)P.n_elem * P.NB);
            for (int e = 0; e < P.n_elem; e++)
                for (int k = 0; k < P.NB; k++) rows.back()[(size_t)e * P.NB + k] = (int32_t)row_of(P, e, k);
            tables.push_back(ElemTable{rows.back().data(), P.n_elem, P.NB});
        }
    graph_partition_rows(nbr, c.world, tables, hub.data(), owner);
}
}  // namespace

// block rows in breadth-first order of the graph of the potentials with fixed host connectivity (hubs — rows of small DoF sets — and rows no
// element touches at the end): the solver numbering of one GPU when the caller handed over no positions (kernels.hip: prepare)
void static_graph_order(Context& c, std::vector<int32_t>& rows_in_order)
{
    const int64_t nbr = c.nbr;
    std::vector<uint8_t> hub((size_t)nbr, 0);
    for (auto& s : c.dof_sets) {
        const int64_t rows = s.n / 3;
        if (rows > 0 && rows <= HOT_SET_ROWS)
            for (int64_t r = 0; r < rows; r++) hub[(size_t)(s.offset / 3 + r)] = 1;
    }
    std::vector<std::vector<int32_t>> rows;
    std::vector<ElemTable> tables;
    for (auto& P : c.pots)
        if (P.part == 0 && has_host_conn(P)) {
            rows.emplace_back((size_t)P.n_elem * P.NB);
            for (int e = 0; e < P.n_elem; e++)
                for (int k = 0; k < P.NB; k++) rows.back()[(size_t)e * P.NB + k] = (int32_t)row_of(P, e, k);
            tables.push_back(ElemTable{rows.back().data(), P.n_elem, P.NB});
        }
    std::vector<int64_t> order, weight;
    graph_bfs_order(nbr, tables, hub.data(), order, weight);
    rows_in_order.clear();
    std::vector<uint8_t> seen((size_t)nbr, 0);
    for (int64_t v : order) {
        rows_in_order.push_back((int32_t)v);
        seen[(size_t)v] = 1;
    }
    for (int64_t r = 0; r < nbr; r++)
        if (!seen[(size_t)r]) rows_in_order.push_back((int32_t)r);
}

void shard_prepare(Context& c)
{
    Shard& S = c.sh;
    const int W = c.world, me = c.rank;
    const int64_t nbr = c.nbr;
    if (W > 30) throw Error("sharded runs support up to 30 ranks");
    // ---- what the partition and the lists depend on
    std::vector<int32_t> shared = S.shared_rows;
    contact_shared_rows(c, shared);
    auto hash_bytes = [](const void* p, size_t n) {  // (contents, not only sizes: shared rows that change at equal count must refresh the ghosts)
        const unsigned char* b = static_cast<const unsigned char*>(p);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
        return (int64_t)h;
    };
    std::vector<int64_t> sig{nbr, (int64_t)W, (int64_t)me, (int64_t)S.user_owner.size(), (int64_t)S.coords.size(), (int64_t)shared.size(), (int64_t)S.version,
                             hash_bytes(shared.data(), shared.size() * sizeof(int32_t))};
    for (auto& P : c.pots) {
        sig.push_back(shards_by_list(P) ? (int64_t)P.conn_version : -1);
        sig.push_back(P.part);
        sig.push_back(shards_by_list(P) ? P.n_elem : -1);
        for (int k = 0; k < P.NB; k++) {
            sig.push_back(P.args.dof_col[k]);
            sig.push_back(P.args.dof_row_off[k]);
        }
    }
    if (sig != S.sig) {
        S.sig = sig;
        if (!S.user_owner.empty()) {
            if ((int64_t)S.user_owner.size() != nbr) throw Error("mistark_dist_set_row_owner: the map must have one entry per block row (" + std::to_string(nbr) + ")");
            for (int32_t o : S.user_owner)
                if (o < 0 || o >= W) throw Error("mistark_dist_set_row_owner: owner out of range");
            S.owner = S.user_owner;
        } else if (!S.coords.empty()) {
            if ((int64_t)S.coords.size() != 3 * nbr) throw Error("mistark_dist_set_row_coords: one position per block row (" + std::to_string(nbr) + ")");
            std::vector<int64_t> weight((size_t)nbr, 1);
            for (auto& P : c.pots)
                if (shards_by_list(P))
                    for (int e = 0; e < P.n_elem; e++)
                        for (int k = 0; k < P.NB; k++) weight[(size_t)row_of(P, e, k)]++;
            rcb_partition_rows(nbr, W, S.coords.data(), weight, S.owner);
        } else {
            graph_partition(c, S.owner);
        }
        // ---- which ranks need which rows (bit r: rank r evaluates an element touching the row)
        std::vector<uint32_t> need((size_t)nbr, 0u);
        const uint32_t all = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
        for (auto& P : c.pots) {
            if (!shards_by_list(P)) continue;
            for (int e = 0; e < P.n_elem; e++) {
                uint32_t m = 0;
                for (int k = 0; k < P.NB; k++) m |= 1u << S.owner[(size_t)row_of(P, e, k)];
                for (int k = 0; k < P.NB; k++) need[(size_t)row_of(P, e, k)] |= m;
            }
        }
        for (int32_t r : shared) {
            if (r < 0 || r >= nbr) throw Error("mistark_dist_add_shared_rows: row out of range");
            need[(size_t)r] = all;
        }
        for (auto& s : c.dof_sets) {  // rigid bodies and other small sets: every contact / attachment may reference them
            const int64_t rows = s.n / 3;
            if (rows > 0 && rows <= HOT_SET_ROWS)
                for (int64_t r = 0; r < rows; r++) need[(size_t)(s.offset / 3 + r)] = all;
        }
        // ---- send rows of every rank and my local numbering
        S.n_own_of.assign((size_t)W, 0);
        S.n_send_of.assign((size_t)W, 0);
        std::vector<int32_t> pos_in_send((size_t)nbr, -1), local_of((size_t)nbr, -1);
        for (int64_t r = 0; r < nbr; r++) {
            const int o = S.owner[(size_t)r];
            local_of[(size_t)r] = (int32_t)S.n_own_of[(size_t)o]++;  // (local index on the owner)
            if (need[(size_t)r] & ~(1u << o)) pos_in_send[(size_t)r] = (int32_t)S.n_send_of[(size_t)o]++;
        }
        S.n_own = S.n_own_of[(size_t)me];
        S.n_send = S.n_send_of[(size_t)me];
        S.send_stride = std::max<int64_t>(1, *std::max_element(S.n_send_of.begin(), S.n_send_of.end()));
        S.own_stride = std::max<int64_t>(1, *std::max_element(S.n_own_of.begin(), S.n_own_of.end()));
        S.grow_h.clear();
        std::vector<int32_t> send_rows, send_pos_of_row;
        std::vector<uint32_t> send_mask;
        for (int64_t r = 0; r < nbr; r++)
            if (S.owner[(size_t)r] == me) {
                S.grow_h.push_back((int32_t)r);
                send_pos_of_row.push_back(pos_in_send[(size_t)r]);
                if (pos_in_send[(size_t)r] >= 0) {
                    send_rows.push_back(local_of[(size_t)r]);
                    send_mask.push_back(need[(size_t)r] & ~(1u << me));
                }
            }
        std::vector<int32_t> ghost_src;
        for (int o = 0; o < W; o++) {
            if (o == me) continue;
            for (int64_t r = 0; r < nbr; r++)
                if (S.owner[(size_t)r] == o && (need[(size_t)r] & (1u << me))) {
                    S.grow_h.push_back((int32_t)r);
                    ghost_src.push_back((int32_t)((int64_t)o * S.send_stride + pos_in_send[(size_t)r]));
                }
        }
        S.n_loc = (int64_t)S.grow_h.size();
        S.n_ghost = S.n_loc - S.n_own;
        std::vector<int32_t> lrow((size_t)nbr, -1);
        for (int64_t i = 0; i < S.n_loc; i++) lrow[(size_t)S.grow_h[(size_t)i]] = (int32_t)i;
        std::vector<int32_t> grow_all((size_t)W * (size_t)S.own_stride, -1);
        for (int64_t r = 0; r < nbr; r++) grow_all[(size_t)S.owner[(size_t)r] * (size_t)S.own_stride + (size_t)local_of[(size_t)r]] = (int32_t)r;
        auto up = [&](DevBuf<int32_t>& d, const std::vector<int32_t>& h) {
            d.ensure(std::max<size_t>(h.size(), 1));
            if (!h.empty()) MS_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
        };
        up(S.lrow, lrow);
        up(S.grow, S.grow_h);
        up(S.send_rows, send_rows);
        up(S.ghost_src, ghost_src);
        up(S.send_pos_of_row, send_pos_of_row);
        S.send_mask.ensure(std::max<size_t>(send_mask.size(), 1));
        if (!send_mask.empty()) MS_CHECK(hipMemcpyAsync(S.send_mask.p, send_mask.data(), send_mask.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
        up(S.grow_all, grow_all);
        S.sendbuf.ensure(3 * (size_t)S.send_stride);
        S.recvbuf.ensure(3 * (size_t)S.send_stride * (size_t)W);
        S.gath_s.ensure(3 * (size_t)S.own_stride);
        S.gath_r.ensure(3 * (size_t)S.own_stride * (size_t)W);
        S.scal_s.ensure(64);
        S.scal_r.ensure(64 * (size_t)W);
        S.err.ensure(1);
        MS_CHECK(hipMemsetAsync(S.err.p, 0, sizeof(int32_t), c.stream));
        // ---- element lists: [energy counts here | interface elements of other ranks]
        for (auto& P : c.pots) {
            P.n_list = P.n_eown_list = 0;
            if (!shards_by_list(P)) continue;
            std::vector<uint32_t> mine, halo;
            for (int e = 0; e < P.n_elem; e++) {
                bool touch = false;
                for (int k = 0; k < P.NB; k++) touch = touch || S.owner[(size_t)row_of(P, e, k)] == me;
                if (!touch) continue;
                (S.owner[(size_t)row_of(P, e, 0)] == me ? mine : halo).push_back((uint32_t)e);
            }
            P.n_eown_list = (int)mine.size();
            mine.insert(mine.end(), halo.begin(), halo.end());
            P.n_list = (int)mine.size();
            P.elem_list.ensure(std::max<size_t>(mine.size(), 1));
            if (!mine.empty()) MS_CHECK(hipMemcpyAsync(P.elem_list.p, mine.data(), mine.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
        }
        MS_CHECK(hipStreamSynchronize(c.stream));  // (host vectors above are temporaries)
        c.part[0].dirty = c.part[1].dirty = true;
        S.version_lists++;
    }
    // ---- kernel argument blocks
    for (auto& P : c.pots) {
        PotArgs& A = P.args;
        A.lrow = S.lrow.p;
        A.n_own = (int)S.n_own;
        if (shards_by_list(P)) {
            A.elem_list = P.elem_list.p;
            A.e_begin = 0;
            A.e_count = P.n_list;
            P.n_key = P.n_list;
            P.n_eown = P.n_eown_list;
        } else {  // connectivity written on the device (contact tables): small, evaluated by every rank; rows of other ranks are dropped
            A.elem_list = nullptr;
            A.e_begin = 0;
            A.e_count = P.n_elem;
            P.n_key = P.n_elem;
            P.n_eown = P.n_elem;
        }
        A.n_pool = P.n_key;
    }
}

void shard_check(Context& c)
{
    int32_t e = 0;
    fetch(c, &e, c.sh.err.p, sizeof(int32_t));
    if (e) throw Error("sharded run: a potential with device-side connectivity references a block row that is neither owned by this rank nor registered as shared (mistark_dist_add_shared_rows)");
}

static void halo_impl(Context& c, double* v, bool global)
{
    Shard& S = c.sh;
    if (S.n_send > 0)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid_of(3 * S.n_send)), dim3(TB), 0, c.stream, (const double*)v, (const int32_t*)S.send_rows.p, global ? (const int32_t*)S.grow.p : nullptr,
                           S.n_send, S.sendbuf.p);
    c.coll->allgather_f64(S.sendbuf.p, S.recvbuf.p, 3 * (size_t)S.send_stride, c.stream);
    if (S.n_ghost > 0)
        hipLaunchKernelGGL(k_unpack_ghosts, dim3(grid_of(3 * S.n_ghost)), dim3(TB), 0, c.stream, (const double*)S.recvbuf.p, (const int32_t*)S.ghost_src.p, S.n_ghost, S.n_own,
                           global ? (const int32_t*)S.grow.p : nullptr, v);
}
void shard_halo(Context& c, double* v_local) { halo_impl(c, v_local, false); }
void shard_halo_global(Context& c, double* v_global) { halo_impl(c, v_global, true); }

void shard_to_local(Context& c, const double* v_global, double* v_local, bool with_ghosts)
{
    const int64_t n = with_ghosts ? c.sh.n_loc : c.sh.n_own;
    if (n > 0) hipLaunchKernelGGL(k_to_local, dim3(grid_of(3 * n)), dim3(TB), 0, c.stream, v_global, (const int32_t*)c.sh.grow.p, n, v_local);
}
void shard_gather_global(Context& c, const double* v_local, double* v_global)
{
    Shard& S = c.sh;
    hipLaunchKernelGGL(k_copy_pad, dim3(grid_of(3 * S.own_stride)), dim3(TB), 0, c.stream, v_local, 3 * S.n_own, 3 * S.own_stride, S.gath_s.p);
    c.coll->allgather_f64(S.gath_s.p, S.gath_r.p, 3 * (size_t)S.own_stride, c.stream);
    const int64_t n = S.own_stride * c.world;
    hipLaunchKernelGGL(k_scatter_all, dim3(grid_of(3 * n)), dim3(TB), 0, c.stream, (const double*)S.gath_r.p, (const int32_t*)S.grow_all.p, n, v_global);
}
void shard_allgather_scalars(Context& c, const double* mine, int n, double* all_host)
{
    Shard& S = c.sh;
    if (n > 64) throw Error("shard_allgather_scalars: too many values");
    MS_CHECK(hipMemcpyAsync(S.scal_s.p, mine, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));  // (`mine` is the caller's stack)
    c.coll->allgather_f64(S.scal_s.p, S.scal_r.p, (size_t)n, c.stream);
    fetch(c, all_host, S.scal_r.p, (size_t)n * (size_t)c.world * sizeof(double));
}
double shard_sum(Context& c, double mine)
{
    double all[64];
    shard_allgather_scalars(c, &mine, 1, all);
    double s = 0.0;
    for (int r = 0; r < c.world; r++) s += all[r];  // rank order: the same bits on every rank
    return s;
}

}  // namespace mistark
