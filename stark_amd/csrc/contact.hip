// contact.hip — device contact detector: collision vertices, candidate search, closest-feature classification, routing into the
// 21 barrier and 14 friction tables of stark::EnergyFrictionalContact, edge-triangle intersection count.
//
// Reference behaviour restated (see include/mistark_contact.h for the API mapping):
//   vertices        EnergyFrictionalContact::_update_vertices                 EnergyFrictionalContact.cpp:219-250
//   candidates      tmcd broad phase: enlarged float AABBs, orphan/blacklist   BroadPhasePTEEBase.cpp:162-270, AABBs.cpp:15-45
//   narrow phase    tmcd::ProximityDetection::run                             ProximityDetection.cpp:75-190 (contact_geom.hpp)
//   routing         _before_energy_evaluation__update_contacts                EnergyFrictionalContact.cpp:368-530
//   friction        _before_time_step__update_friction_contacts               EnergyFrictionalContact.cpp:531-773
//   intersections   tmcd::IntersectionDetection::run                          IntersectionDetection.cpp:50-95, BroadPhaseET.cpp:161-165
//
// Pipeline of one update (all on the engine's stream):
//   k_contact_vertices -> k_contact_aabbs -> k_detect_pt / k_detect_ee (LDS-tiled all-pairs AABB rejection, narrow phase on the
//   survivors, one 64-bit key per contact = table | source | closest feature | primitive a | primitive b) -> radix sort of the
//   keys (deterministic row order, tables contiguous) -> k_table_bounds -> row counts to the host -> k_route writes every row
//   (and the friction data) into the buffers the potentials' kernels read.
// If the sorted key list equals the previous one the tables are left alone and the dynamic matrix pattern is kept.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <hipcub/hipcub.hpp>
#include <map>
#include <string>
#include <vector>

#include "../../include/mistark_contact.h"
#include "contact_geom.hpp"
#include "energies.hpp"
#include "engine.hpp"
#include "dist.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
namespace mistark {

namespace {
struct HostProf
{
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long n = 0;
    bool on = getenv("MISTARK_PROF_CONTACT") != nullptr;
    ~HostProf()
    {
        if (on && n) std::fprintf(stderr, "contact host profile over %ld updates (ms): prep %.2f verts+sort+sweep launch %.2f fetch1 %.2f keysort+bounds launch %.2f fetch2 %.2f install %.2f route %.2f\n", n, 1e3 * t[0], 1e3 * t[1], 1e3 * t[2], 1e3 * t[3], 1e3 * t[4], 1e3 * t[5], 1e3 * t[6]);
    }
};
HostProf g_prof;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int CB = 256;  // workgroup size of the detector kernels
constexpr int N_TABLES = 35;
constexpr int N_CONTACT_TABLES = 21;
enum Role { R_V1 = 0, R_X0, R_X, R_DT, R_K, R_THICK, R_EPSV, R_RB_XLOC, R_RB_V1, R_RB_W1, R_RB_T0, R_RB_Q0, R_T, R_MU, R_FN, R_BARY };
const char* const TABLE_NAMES[N_TABLES] = {
    "contact_d_d_pt_pp_cubic", "contact_d_d_pt_pe_cubic", "contact_d_d_pt_pt_cubic", "contact_d_d_ee_pp_cubic", "contact_d_d_ee_pe_cubic", "contact_d_d_ee_ee_cubic",
    "contact_rb_rb_pt_pp_cubic", "contact_rb_rb_pt_pe_cubic", "contact_rb_rb_pt_pt_cubic", "contact_rb_rb_ee_pp_cubic", "contact_rb_rb_ee_pe_cubic", "contact_rb_rb_ee_ee_cubic",
    "contact_rb_d_pt_pp_cubic", "contact_rb_d_pt_pe_cubic", "contact_rb_d_pt_pt_cubic", "contact_rb_d_pt_ep_cubic", "contact_rb_d_pt_tp_cubic",
    "contact_rb_d_ee_pp_cubic", "contact_rb_d_ee_pe_cubic", "contact_rb_d_ee_ee_cubic", "contact_rb_d_ee_ep_cubic",
    "friction_d_d_pp_C0", "friction_d_d_pe_C0", "friction_d_d_pt_C0", "friction_d_d_ee_C0",
    "friction_rb_rb_pp_C0", "friction_rb_rb_pe_C0", "friction_rb_rb_pt_C0", "friction_rb_rb_ee_C0",
    "friction_rb_d_pp_C0", "friction_rb_d_pe_C0", "friction_rb_d_pt_C0", "friction_rb_d_ee_C0", "friction_rb_d_ep_C0", "friction_rb_d_tp_C0"};

struct Bind
{
    int role, stride, col;
};
// Binding recipe of table t: the reference's mws.make_* calls in order (EnergyFrictionalContact.cpp:829-1218 with the symbol
// getters of :1358-1423). Systems: 0 = deformable, 1 = rigid body.
std::vector<Bind> recipe(int t, int& conn_stride)
{
    std::vector<Bind> r;
    auto d_x1 = [&](int c0, int n) {  // _get_d_x1
        for (int i = 0; i < n; i++) r.push_back({R_V1, 3, c0 + i});
        for (int i = 0; i < n; i++) r.push_back({R_X0, 3, c0 + i});
        r.push_back({R_DT, 1, -1});
    };
    auto d_X = [&](int c0, int n) { for (int i = 0; i < n; i++) r.push_back({R_X, 3, c0 + i}); };
    auto d_v1 = [&](int c0, int n) { for (int i = 0; i < n; i++) r.push_back({R_V1, 3, c0 + i}); };
    auto rb = [&](int rbcol, int c0, int n) {  // _get_rb_x1 / _get_rb_v1
        r.push_back({R_DT, 1, -1});
        for (int i = 0; i < n; i++) r.push_back({R_RB_XLOC, 3, c0 + i});
        r.push_back({R_RB_V1, 3, rbcol});
        r.push_back({R_RB_W1, 3, rbcol});
        r.push_back({R_RB_T0, 3, rbcol});
        r.push_back({R_RB_Q0, 4, rbcol});
    };
    auto rb_X = [&](int c0, int n) { for (int i = 0; i < n; i++) r.push_back({R_RB_XLOC, 3, c0 + i}); };
    if (t < N_CONTACT_TABLES) {
        const int fam_sys = t < 6 ? 0 : (t < 12 ? 1 : 2);  // d_d, rb_rb, rb_d
        const int sysA = fam_sys == 0 ? 0 : 1, sysB = fam_sys == 1 ? 1 : 0;
        const int base = fam_sys == 0 ? 2 : (fam_sys == 1 ? 4 : 3);
        const int rbA = 2, rbB = fam_sys == 1 ? 3 : 2;
        const int k = fam_sys == 0 ? t : (fam_sys == 1 ? t - 6 : t - 12);
        auto pos = [&](int sys, int rbcol, int c0, int n) { sys == 0 ? d_x1(c0, n) : rb(rbcol, c0, n); };
        auto rest = [&](int sys, int c0, int n) { sys == 0 ? d_X(c0, n) : rb_X(c0, n); };
        // point-triangle family (KA, KB) / edge-edge family (point on A, point on B)
        int KA = 0, KB = 0, PA = -1, PB = -1;
        if (fam_sys < 2) {
            static const int ka[6] = {1, 1, 1, 0, 0, 0}, kb[6] = {1, 2, 3, 0, 0, 0}, pa[6] = {-1, -1, -1, 1, 1, 0}, pb[6] = {-1, -1, -1, 1, 0, 0};
            KA = ka[k]; KB = kb[k]; PA = pa[k]; PB = pb[k];
        } else {
            static const int ka[9] = {1, 1, 1, 2, 3, 0, 0, 0, 0}, kb[9] = {1, 2, 3, 1, 1, 0, 0, 0, 0}, pa[9] = {-1, -1, -1, -1, -1, 1, 1, 0, 0}, pb[9] = {-1, -1, -1, -1, -1, 1, 0, 0, 1};
            KA = ka[k]; KB = kb[k]; PA = pa[k]; PB = pb[k];
        }
        if (PA < 0) {
            pos(sysA, rbA, base, KA);
            pos(sysB, rbB, base + KA, KB);
            conn_stride = base + KA + KB;
            r.push_back({R_THICK, 1, 0});
            r.push_back({R_THICK, 1, 1});
            r.push_back({R_K, 1, -1});
        } else {
            int col = base;
            for (int side = 0; side < 2; side++) {
                const int sys = side == 0 ? sysA : sysB, rbcol = side == 0 ? rbA : rbB, has_p = side == 0 ? PA : PB;
                pos(sys, rbcol, col, 2);
                rest(sys, col, 2);
                col += 2;
                if (has_p) {
                    pos(sys, rbcol, col, 1);
                    col += 1;
                }
            }
            conn_stride = col;
            r.push_back({R_K, 1, -1});
            r.push_back({R_THICK, 1, 0});
            r.push_back({R_THICK, 1, 1});
        }
    } else {
        const int f = t - N_CONTACT_TABLES;
        const int fam_sys = f < 4 ? 0 : (f < 8 ? 1 : 2);
        const int k = fam_sys == 0 ? f : (fam_sys == 1 ? f - 4 : f - 8);  // pp pe pt ee [ep tp]
        static const int ka[6] = {1, 1, 1, 2, 2, 3}, kb[6] = {1, 2, 3, 2, 1, 1}, nb[6] = {0, 2, 3, 2, 2, 3};
        const int base = fam_sys == 0 ? 1 : (fam_sys == 1 ? 3 : 2);
        const int sysA = fam_sys == 0 ? 0 : 1, sysB = fam_sys == 1 ? 1 : 0;
        sysA == 0 ? d_v1(base, ka[k]) : rb(1, base, ka[k]);
        sysB == 0 ? d_v1(base + ka[k], kb[k]) : rb(2, base + ka[k], kb[k]);
        conn_stride = base + ka[k] + kb[k];
        if (nb[k]) r.push_back({R_BARY, nb[k], 0});
        r.push_back({R_T, 6, 0});
        r.push_back({R_MU, 1, 0});
        r.push_back({R_FN, 1, 0});
        r.push_back({R_EPSV, 1, -1});
        r.push_back({R_DT, 1, -1});
    }
    return r;
}
int table_nbary(int t)
{
    if (t < N_CONTACT_TABLES) return 0;
    const int f = t - N_CONTACT_TABLES;
    const int k = f < 4 ? f : (f < 8 ? f - 4 : f - 8);
    static const int nb[6] = {0, 2, 3, 2, 2, 3};
    return nb[k];
}

// ---- device-side view of the collision meshes ----------------------------------------------------------------------------------------
struct ContactDev
{
    const int32_t* cv_src;     // collision vertex -> index in the physical system's vertex array
    const int32_t* cv_mesh;    // collision vertex -> mesh (group)
    const int32_t* tri;        // 3 collision vertices per triangle
    const int32_t* tri_mesh;
    const int32_t* edge;       // 2 collision vertices per edge
    const int32_t* edge_mesh;
    const int32_t* mesh_kind;  // 0 deformable, 1 rigid body
    const int32_t* mesh_idx;   // idx_in_ps
    const uint8_t* disabled;   // n_mesh x n_mesh
    const double* mu;          // n_mesh x n_mesh
    const double* thick;       // per mesh
    const double* X;           // 3 per collision vertex
    const float* aabb;         // 6 per primitive: points | triangles | edges
    int n_mesh, n_v, n_t, n_e;
    // range blacklists (tmcd add_blacklist_range_point_triangle / _edge_edge, BroadPhasePTEEBase.cpp:19-41,176-262): half-open intervals of
    // GLOBAL primitive indices, 4 ints each — {triangle begin, end, point begin, end} and {lower edge begin, end, higher edge begin, end}
    const int32_t* bl_pt;
    const int32_t* bl_ee;
    int n_bl_pt, n_bl_ee;
    // broad_only (mistark_cd_run_broad_phase): the sweep emits the candidate pairs themselves — those whose boxes AS THE REFERENCE BUILDS THEM
    // overlap (broad_extra = its enlargement, AABBs.cpp:38) — instead of classifying them
    int broad_only;
    float broad_extra;
};
struct TableDev
{
    int32_t* conn;
    double *T, *mu, *fn, *bary;
    int stride, nbary, start, pad;
};

// key = table[63:58] | source[57] (0 point-triangle, 1 edge-edge) | closest feature[56:53] | primitive a[51:26] | primitive b[25:0]
constexpr int PRIM_BITS = 26;
__host__ __device__ inline uint64_t pack_key(int table, int src, int type, int a, int b)
{
    return ((uint64_t)table << 58) | ((uint64_t)src << 57) | ((uint64_t)type << 53) | ((uint64_t)a << PRIM_BITS) | (uint64_t)b;
}

__device__ __forceinline__ D3 ldx(const double* X, int i) { return d3(X[3 * i], X[3 * i + 1], X[3 * i + 2]); }

// table of a classified pair: family 0 pt_pp 1 pt_pe 2 pt_pt 3 ee_pp 4 ee_pe 5 ee_ee; ka / kb = system of the reference's A / B
__device__ __forceinline__ int table_of(int fam, int ka, int kb, bool friction)
{
    // (written with selects instead of lookup arrays: run-time indexed local arrays live in scratch memory)
    if (!friction) {
        if (ka == 0 && kb == 0) return fam;
        if (ka == 1 && kb == 1) return 6 + fam;
        if (ka == 1) return fam < 3 ? 12 + fam : 14 + fam;  // pt_pp pt_pe pt_pt | ee_pp ee_pe ee_ee -> 12 13 14 | 17 18 19
        // deformable first: the rigid side is listed first and the asymmetric families switch table: 12 15 16 | 17 20 19
        return fam == 0 ? 12 : (fam == 1 ? 15 : (fam == 2 ? 16 : (fam == 3 ? 17 : (fam == 4 ? 20 : 19))));
    }
    const int k4 = fam < 3 ? fam : (fam == 5 ? 3 : fam - 3);  // pp pe pt | pp pe ee
    if (ka == 0 && kb == 0) return 21 + k4;
    if (ka == 1 && kb == 1) return 25 + k4;
    if (ka == 1) return 29 + k4;
    return k4 == 0 ? 29 : (k4 == 1 ? 33 : (k4 == 2 ? 34 : 32));  // pp -> pp, pe -> ep, pt -> tp, ee -> ee
}
__device__ __forceinline__ int pt_family(int type) { return type <= P_T2 ? 0 : (type <= P_E2 ? 1 : 2); }
__device__ __forceinline__ int ee_family(int type) { return type <= EA1_EB1 ? 3 : (type <= EA1_EB ? 4 : 5); }

// ---- collision vertices ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CB) void k_contact_vertices(ContactDev d, const double* __restrict__ x0, const double* __restrict__ v1, const double* __restrict__ xloc,
                                                        const double* __restrict__ rb_v1, const double* __restrict__ rb_w1, const double* __restrict__ rb_t0,
                                                        const double* __restrict__ rb_q0, double dt, double* __restrict__ X)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    if (i >= d.n_v) return;
    const int m = d.cv_mesh[i], s = d.cv_src[i];
    if (d.mesh_kind[m] == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) X[3 * i + k] = x0[3 * s + k] + dt * v1[3 * s + k];
    } else {
        // integrate_loc_point: t1 = t0 + dt v1, R1 = R(normalize(q0 + dt/2 (0, w1) q0)) (rigidbody_transformations.cpp:30-43)
        const int b = d.mesh_idx[m];
        const V3<double> w(rb_w1[3 * b], rb_w1[3 * b + 1], rb_w1[3 * b + 2]);
        const M3<double> R1 = rb_R1(rb_q0 + 4 * b, w, dt);
        const V3<double> xl(xloc[3 * s], xloc[3 * s + 1], xloc[3 * s + 2]);
        const V3<double> r = R1 * xl;
        X[3 * i] = r.x + (rb_t0[3 * b] + dt * rb_v1[3 * b]);
        X[3 * i + 1] = r.y + (rb_t0[3 * b + 1] + dt * rb_v1[3 * b + 1]);
        X[3 * i + 2] = r.z + (rb_t0[3 * b + 2] + dt * rb_v1[3 * b + 2]);
    }
}
// float AABBs rounded outwards and enlarged (AABBs.cpp:15-45); primitive order: points, triangles, edges
__global__ __launch_bounds__(CB) void k_contact_aabbs(ContactDev d, float enl, float* __restrict__ aabb)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    const int n = d.n_v + d.n_t + d.n_e;
    if (i >= n) return;
    int v[3], nv;
    if (i < d.n_v) {
        v[0] = i;
        nv = 1;
    } else if (i < d.n_v + d.n_t) {
        const int t = i - d.n_v;
        v[0] = d.tri[3 * t]; v[1] = d.tri[3 * t + 1]; v[2] = d.tri[3 * t + 2];
        nv = 3;
    } else {
        const int e = i - d.n_v - d.n_t;
        v[0] = d.edge[2 * e]; v[1] = d.edge[2 * e + 1];
        nv = 2;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double lo = d.X[3 * v[0] + k], hi = lo;
        for (int j = 1; j < nv; j++) {
            const double x = d.X[3 * v[j] + k];
            lo = x < lo ? x : lo;
            hi = x > hi ? x : hi;
        }
        aabb[6 * (size_t)i + k] = __double2float_rd(lo) - enl;
        aabb[6 * (size_t)i + 3 + k] = __double2float_ru(hi) + enl;
    }
}

// ---- candidate search + narrow phase --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void push_key(uint64_t key, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    const int slot = atomicAdd(&counters[0], 1);
    if (slot < key_cap) keys[slot] = key;
}
// The reference's own box of a primitive along one axis (AABBs.cpp:7-21): the vertices cast to float (to nearest), their minimum / maximum, minus /
// plus the float enlargement. The search's boxes (k_contact_aabbs) are rounded outwards and a superset of these; a broad-phase LISTING has to
// hold exactly the pairs the reference's overlap test (BroadPhasePTEEBase.cpp: <= both ways on every axis) accepts.
__device__ __forceinline__ void ref_box(const double* X, const int* v, int nv, int k, float extra, float& lo, float& hi)
{
    lo = hi = __double2float_rn(X[3 * v[0] + k]);
    for (int j = 1; j < nv; j++) {
        const float x = __double2float_rn(X[3 * v[j] + k]);
        lo = fminf(lo, x);
        hi = fmaxf(hi, x);
    }
    lo = __fsub_rn(lo, extra);
    hi = __fadd_rn(hi, extra);
}
__device__ __forceinline__ bool ref_boxes_overlap(const ContactDev& d, const int* va, int na, const int* vb, int nb)
{
    for (int k = 0; k < 3; k++) {
        float alo, ahi, blo, bhi;
        ref_box(d.X, va, na, k, d.broad_extra, alo, ahi);
        ref_box(d.X, vb, nb, k, d.broad_extra, blo, bhi);
        if (!(alo <= bhi && blo <= ahi)) return false;
    }
    return true;
}
template <bool FRICTION>
__device__ __forceinline__ void narrow_pt(const ContactDev& d, int p, int t, double enl2, uint64_t* keys, int* counters, int key_cap)
{
    const int v0 = d.tri[3 * t], v1 = d.tri[3 * t + 1], v2 = d.tri[3 * t + 2];
    if (p == v0 || p == v1 || p == v2) return;  // point of its own triangle (BroadPhasePTEEBase.cpp:193)
    const int mp = d.cv_mesh[p], mt = d.tri_mesh[t];
    if (d.disabled[mp * d.n_mesh + mt]) return;
    for (int k = 0; k < d.n_bl_pt; k++)  // BroadPhasePTEEBase.cpp:181-205
        if (d.bl_pt[4 * k] <= t && t < d.bl_pt[4 * k + 1] && d.bl_pt[4 * k + 2] <= p && p < d.bl_pt[4 * k + 3]) return;
    if (d.broad_only) {  // BroadPhasePTEEBase.cpp:190-214: the pair itself
        const int vt[3] = {v0, v1, v2};
        if (ref_boxes_overlap(d, &p, 1, vt, 3)) push_key(pack_key(0, 0, 0, p, t), keys, counters, key_cap);
        return;
    }
    int type;
    const double d2 = point_triangle_sq_distance(type, ldx(d.X, p), ldx(d.X, v0), ldx(d.X, v1), ldx(d.X, v2));
    if (!(d2 < enl2)) return;                                           // ProximityDetection.cpp:105
    if (::sqrt(d2) > d.thick[mp] + d.thick[mt]) return;                  // EnergyFrictionalContact.cpp:385
    if (FRICTION && d.mu[mp * d.n_mesh + mt] == 0.0) return;            // :597
    const int table = table_of(pt_family(type), d.mesh_kind[mp], d.mesh_kind[mt], FRICTION);
    push_key(pack_key(table, 0, type, p, t), keys, counters, key_cap);
}
template <bool FRICTION>
__device__ __forceinline__ void narrow_ee(const ContactDev& d, int ea, int eb, double enl2, uint64_t* keys, int* counters, int key_cap)
{
    const int a0 = d.edge[2 * ea], a1 = d.edge[2 * ea + 1], b0 = d.edge[2 * eb], b1 = d.edge[2 * eb + 1];
    if (a0 == b0 || a0 == b1 || a1 == b0 || a1 == b1) return;  // edges sharing a vertex (BroadPhasePTEEBase.cpp:246)
    const int ma = d.edge_mesh[ea], mb = d.edge_mesh[eb];
    if (d.disabled[ma * d.n_mesh + mb]) return;
    for (int k = 0; k < d.n_bl_ee; k++) {  // (the pair is looked up as (lower, higher) global edge: BroadPhasePTEEBase.cpp:229-258)
        const int lo = ea < eb ? ea : eb, hi = ea < eb ? eb : ea;
        if (d.bl_ee[4 * k] <= lo && lo < d.bl_ee[4 * k + 1] && d.bl_ee[4 * k + 2] <= hi && hi < d.bl_ee[4 * k + 3]) return;
    }
    if (d.broad_only) {  // BroadPhasePTEEBase.cpp:236-262 (the parallel-edge cutoff belongs to the narrow phase)
        const int va[2] = {a0, a1}, vb[2] = {b0, b1};
        if (ref_boxes_overlap(d, va, 2, vb, 2)) push_key(pack_key(1, 1, 0, ea < eb ? ea : eb, ea < eb ? eb : ea), keys, counters, key_cap);
        return;
    }
    const D3 xa0 = ldx(d.X, a0), xa1 = ldx(d.X, a1), xb0 = ldx(d.X, b0), xb1 = ldx(d.X, b1);
    if (sq3(cross3(xa1 - xa0, xb1 - xb0)) <= 1e-30) return;  // (nearly) parallel edges never reach a table (ProximityDetection.cpp:152-155)
    int type;
    const double d2 = edge_edge_sq_distance(type, xa0, xa1, xb0, xb1);
    if (!(d2 < enl2)) return;
    if (::sqrt(d2) > d.thick[ma] + d.thick[mb]) return;
    if (FRICTION && d.mu[ma * d.n_mesh + mb] == 0.0) return;
    // roles: the reference's "first" is the edge-point; for EA_EB0 / EA_EB1 that is edge b (ProximityDetection.cpp:171-172)
    const bool swapped = type == EA_EB0 || type == EA_EB1;
    const int ka = d.mesh_kind[swapped ? mb : ma], kb = d.mesh_kind[swapped ? ma : mb];
    const int table = table_of(ee_family(type), ka, kb, FRICTION);
    push_key(pack_key(table, 1, type, ea, eb), keys, counters, key_cap);
}
__device__ __forceinline__ bool overlap6(const float* a, const float* lo, const float* hi, int j)
{
    return a[0] <= hi[j] && lo[j] <= a[3] && a[1] <= hi[CB + j] && lo[CB + j] <= a[4] && a[2] <= hi[2 * CB + j] && lo[2 * CB + j] <= a[5];
}
// All point x triangle pairs: blockIdx.x = tile of CB points (one per lane), blockIdx.y = chunk of triangles staged through LDS
template <bool FRICTION>
__global__ __launch_bounds__(CB) void k_detect_pt(ContactDev d, double enl2, int chunk, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    __shared__ float s_lo[3 * CB], s_hi[3 * CB];
    const int p = blockIdx.x * CB + threadIdx.x;
    const bool valid = p < d.n_v;
    float a[6];
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = valid ? d.aabb[6 * (size_t)p + k] : 0.f;
    const int t_begin = blockIdx.y * chunk;
    const int t_end = min(d.n_t, t_begin + chunk);
    for (int base = t_begin; base < t_end; base += CB) {
        const int tj = base + threadIdx.x;
        if (tj < t_end) {
            const float* bb = d.aabb + 6 * (size_t)(d.n_v + tj);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s_lo[k * CB + threadIdx.x] = bb[k];
                s_hi[k * CB + threadIdx.x] = bb[3 + k];
            }
        }
        __syncthreads();
        if (valid) {
            const int cnt = min(CB, t_end - base);
            for (int j = 0; j < cnt; j++)
                if (overlap6(a, s_lo, s_hi, j)) narrow_pt<FRICTION>(d, p, base + j, enl2, keys, counters, key_cap);
        }
        __syncthreads();
    }
}
// All edge pairs i < j (BroadPhasePTEEBase.cpp:239): row tile x column chunk, tiles entirely below the diagonal exit at once
template <bool FRICTION>
__global__ __launch_bounds__(CB) void k_detect_ee(ContactDev d, double enl2, int chunk, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    __shared__ float s_lo[3 * CB], s_hi[3 * CB];
    const int row0 = blockIdx.x * CB;
    const int e_begin = blockIdx.y * chunk;
    const int e_end = min(d.n_e, e_begin + chunk);
    if (e_end <= row0 + 1) return;  // every column index <= every row index
    const int ea = row0 + threadIdx.x;
    const bool valid = ea < d.n_e;
    const size_t off = (size_t)(d.n_v + d.n_t);
    float a[6];
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = valid ? d.aabb[6 * (off + ea) + k] : 0.f;
    for (int base = e_begin; base < e_end; base += CB) {
        if (base + CB <= row0 + 1) continue;
        const int ej = base + threadIdx.x;
        if (ej < e_end) {
            const float* bb = d.aabb + 6 * (off + ej);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s_lo[k * CB + threadIdx.x] = bb[k];
                s_hi[k * CB + threadIdx.x] = bb[3 + k];
            }
        }
        __syncthreads();
        if (valid) {
            const int cnt = min(CB, e_end - base);
            for (int j = max(0, ea + 1 - base); j < cnt; j++)
                if (overlap6(a, s_lo, s_hi, j)) narrow_ee<FRICTION>(d, ea, base + j, enl2, keys, counters, key_cap);
        }
        __syncthreads();
    }
}
// Edge x triangle intersections (AABBs without enlargement); counters[1] += hits
__global__ __launch_bounds__(CB) void k_detect_et(ContactDev d, int chunk, int* __restrict__ counters)
{
    __shared__ float s_lo[3 * CB], s_hi[3 * CB];
    const int e = blockIdx.x * CB + threadIdx.x;
    const bool valid = e < d.n_e;
    const size_t off = (size_t)(d.n_v + d.n_t);
    float a[6];
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = valid ? d.aabb[6 * (off + e) + k] : 0.f;
    const int t_begin = blockIdx.y * chunk;
    const int t_end = min(d.n_t, t_begin + chunk);
    int hits = 0;
    for (int base = t_begin; base < t_end; base += CB) {
        const int tj = base + threadIdx.x;
        if (tj < t_end) {
            const float* bb = d.aabb + 6 * (size_t)(d.n_v + tj);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s_lo[k * CB + threadIdx.x] = bb[k];
                s_hi[k * CB + threadIdx.x] = bb[3 + k];
            }
        }
        __syncthreads();
        if (valid) {
            const int cnt = min(CB, t_end - base);
            for (int j = 0; j < cnt; j++) {
                if (!overlap6(a, s_lo, s_hi, j)) continue;
                const int t = base + j;
                const int e0 = d.edge[2 * e], e1 = d.edge[2 * e + 1], v0 = d.tri[3 * t], v1 = d.tri[3 * t + 1], v2 = d.tri[3 * t + 2];
                if (e0 == v0 || e0 == v1 || e0 == v2 || e1 == v0 || e1 == v1 || e1 == v2) continue;  // BroadPhaseET.cpp:161-165
                if (d.disabled[d.edge_mesh[e] * d.n_mesh + d.tri_mesh[t]]) continue;
                if (edge_intersects_triangle(ldx(d.X, e0), ldx(d.X, e1), ldx(d.X, v0), ldx(d.X, v1), ldx(d.X, v2))) hits++;
            }
        }
        __syncthreads();
    }
    if (hits) atomicAdd(&counters[1], hits);
}

// ---- banded sweep and prune -----------------------------------------------------------------------------------------------------------
// Boxes are binned into NBANDS slices along a second axis (a box spanning several slices gets one entry per slice) and every
// (class, band) segment is sorted by lo[sweep axis]. A pair of overlapping boxes is reported exactly once: in the band
// max(first band of a, first band of b) — the first band both live in — and there by the box with the smaller lo, inside the
// contiguous run of boxes whose lo lies in [its lo, its hi]. One WAVEFRONT walks one entry's run, 64 candidates per step over a
// sorted copy of the boxes (coalesced), so a box that spans the whole scene (a face of a large rigid body) costs long runs for a few
// waves, not a stalled lane. Same pair set as the all-pairs kernels above (kept as cross-check / ablation).
#ifndef MISTARK_NBANDS
#define MISTARK_NBANDS 64
#endif
constexpr int NBANDS = MISTARK_NBANDS;
constexpr int band_key_bits() { int b = 1; while ((1 << b) <= 3 * NBANDS) b++; return b; }  // (class, band) above the 32-bit sort coordinate; all ones = padding
constexpr int BAND_KEY_BITS = band_key_bits();
static_assert(NBANDS <= (1 << (32 - PRIM_BITS)), "the band of a sorted entry shares its 32-bit index word with the primitive");
constexpr uint32_t SIDX_PRIM = (1u << PRIM_BITS) - 1u;
struct Bands
{
    int axis, band_axis;
    float band_lo, band_scale;
    float shrink;  // intersection sweep over boxes that carry the proximity search's enlargement: twice that enlargement, minus a rounding margin
};
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
}
__device__ __forceinline__ int band_of(const Bands& B, float v)
{
    const float t = (v - B.band_lo) * B.band_scale;
    return t <= 0.f ? 0 : (t >= (float)(NBANDS - 1) ? NBANDS - 1 : (int)t);  // monotone in v, clamped
}
__global__ __launch_bounds__(CB) void k_bp_count(ContactDev d, Bands B, uint32_t* __restrict__ cnt)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    const int n = d.n_v + d.n_t + d.n_e;
    if (i > n) return;
    cnt[i] = i < n ? (uint32_t)(band_of(B, d.aabb[6 * (size_t)i + 3 + B.band_axis]) - band_of(B, d.aabb[6 * (size_t)i + B.band_axis]) + 1) : 0u;
}
__global__ __launch_bounds__(CB) void k_bp_fill(ContactDev d, Bands B, const uint32_t* __restrict__ off, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int cap,
                                               int* __restrict__ counters)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    const int n = d.n_v + d.n_t + d.n_e;
    if (i >= n) return;
    if (i == 0) counters[3] = (int)off[n];  // number of entries (host checks it against the capacity)
    const uint64_t cls = i < d.n_v ? 0 : (i < d.n_v + d.n_t ? 1 : 2);
    const float* b = d.aabb + 6 * (size_t)i;
    const int b0 = band_of(B, b[B.band_axis]), b1 = band_of(B, b[3 + B.band_axis]);
    const uint32_t fk = float_key(b[B.axis]);
    for (int k = b0; k <= b1; k++) {
        const uint32_t e = off[i] + (uint32_t)(k - b0);
        if (e < (uint32_t)cap) {
            keys[e] = ((cls * NBANDS + (uint64_t)k) << 32) | fk;
            idx[e] = (uint32_t)i | ((uint32_t)k << PRIM_BITS);  // (the entry's band rides along: sweep_entry needs it and would otherwise walk the segment starts)
        }
    }
    // the unused tail of the list: padding entries that sort last (a fill command of its own was one more launch in every search)
    for (uint32_t e = off[n] + (uint32_t)i; e < (uint32_t)cap; e += (uint32_t)n) keys[e] = ~0ull;
}
// sorted entries -> sorted copy of the boxes + segment starts seg[cls * NBANDS + band] (seg[3 * NBANDS] = number of entries)
__global__ __launch_bounds__(CB) void k_bp_gather(ContactDev d, Bands B, const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ sidx, int cap,
                                                 float* __restrict__ s_aabb, float* __restrict__ s_lo, int* __restrict__ seg)
{
    const int j = blockIdx.x * CB + threadIdx.x;
    if (j > cap) return;
    const int here = j < cap ? (int)min((uint64_t)(3 * NBANDS), (skeys[j] >> 32) & ((1ull << BAND_KEY_BITS) - 1)) : 3 * NBANDS;  // padding keys sort last
    const int prev = j > 0 ? (int)min((uint64_t)(3 * NBANDS), (skeys[j - 1] >> 32) & ((1ull << BAND_KEY_BITS) - 1)) : -1;
    for (int t = prev + 1; t <= here; t++) seg[t] = j;
    if (j >= cap || here >= 3 * NBANDS) return;
    const float* b = d.aabb + 6 * (size_t)(sidx[j] & SIDX_PRIM);
#pragma unroll
    for (int k = 0; k < 6; k++) s_aabb[6 * (size_t)j + k] = b[k];
    s_lo[j] = b[B.axis];
}
// ---- the sorted box list without a general sort (round 4) ---------------------------------------------------------------------------------
// The list is 3 * NBANDS segments (class, band), each sorted by the lower bound along the sweep axis. The radix sort of 41-bit keys that used to
// produce it was ten launches of the sorting library plus two scans and two fills around it — the longest piece of a search's launch chain (a
// search is ~35 dependent launches of ~5 us whatever their size). Three launches do the same: k_bp_hist counts the entries of every segment
// (LDS histograms, one global atomic per workgroup and touched segment), k_bp_scatter drops every entry {order-preserving lower bound, primitive}
// into its segment (positions from LDS ranks inside a range the workgroup reserved: any order), k_seg_sort sorts each segment in LDS by one
// workgroup quartet (rank sort; the primitive index breaks ties, which is the order the stable radix sort left) and writes the sweep's inputs (sorted
// boxes, lower bounds, index words, segment starts) itself. A segment beyond SEG_SORT_MAX entries raises counters[58]: the host falls back to
// the library sort for good.
// MEASURED (configs[3], 69 k boxes, segments of 700-2100 entries; option "seg_sort"): not a win, OFF by default. The rank sort reads n keys out
// of LDS per entry — a broadcast read still occupies the LDS pipeline for a whole wavefront — 227 us; a bitonic network in LDS with 1024 threads
// 78 us (55-105 barrier-separated steps); the library's radix sort chain ~70 us of kernels. And the search's wait did not move when the sort took
// 78 instead of 70 us in a third of the launches (5.1 against 4.9 ms over 38 searches): the chain is bound by the sweeps (63-89 us + 10-20 us
// each) and the read-back, not by launch count. Kept as a cross-check of the sorted list (identical tables, tests/test_gpu_contact.py).
constexpr int N_SEG = 3 * NBANDS;
constexpr int SEG_SORT_MAX = 8192;
constexpr int SEG_SORT_THREADS = 1024;
__device__ __forceinline__ int seg_of_box(const ContactDev& d, int i) { return i < d.n_v ? 0 : (i < d.n_v + d.n_t ? 1 : 2); }
__global__ __launch_bounds__(CB) void k_bp_hist(ContactDev d, Bands B, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t sh[N_SEG];
    for (int t = threadIdx.x; t < N_SEG; t += CB) sh[t] = 0u;
    __syncthreads();
    const int i = blockIdx.x * CB + threadIdx.x;
    const int n = d.n_v + d.n_t + d.n_e;
    if (i < n) {
        const float* b = d.aabb + 6 * (size_t)i;
        const int b0 = band_of(B, b[B.band_axis]), b1 = band_of(B, b[3 + B.band_axis]), cls = seg_of_box(d, i);
        for (int k = b0; k <= b1; k++) atomicAdd(&sh[cls * NBANDS + k], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < N_SEG; t += CB)
        if (sh[t]) atomicAdd(&hist[t], sh[t]);
}
// hist -> segment starts (every workgroup scans the 192 counts itself); cursor[seg] = entries handed out so far
__global__ __launch_bounds__(CB) void k_bp_scatter(ContactDev d, Bands B, const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor, uint64_t* __restrict__ ekeys, int cap,
                                                  int* __restrict__ counters)
{
    __shared__ uint32_t start[N_SEG + 1], cnt[N_SEG], rank[N_SEG], base[N_SEG];
    // (every thread fetches one count, then sums the counts before its own out of LDS: a single thread walking the 192 counts in global memory
    // was a chain of dependent loads at the head of every workgroup)
    for (int t = threadIdx.x; t < N_SEG; t += CB) {
        base[t] = hist[t];
        cnt[t] = rank[t] = 0u;
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= N_SEG; t += CB) {
        uint32_t at = 0;
        for (int u = 0; u < t; u++) at += base[u];
        start[t] = at;
        if (t == N_SEG && blockIdx.x == 0) counters[3] = (int)at;  // number of entries (the host checks it against the capacity)
    }
    __syncthreads();
    const int i = blockIdx.x * CB + threadIdx.x;
    const int n = d.n_v + d.n_t + d.n_e;
    int b0 = 0, b1 = -1, cls = 0;
    uint32_t fk = 0u;
    if (i < n) {
        const float* b = d.aabb + 6 * (size_t)i;
        b0 = band_of(B, b[B.band_axis]);
        b1 = band_of(B, b[3 + B.band_axis]);
        cls = seg_of_box(d, i);
        fk = float_key(b[B.axis]);
    }
    // the workgroup's entries per segment (LDS atomics), one global reservation per touched segment, then the stores at LDS ranks inside it
    for (int k = b0; k <= b1; k++) atomicAdd(&cnt[cls * NBANDS + k], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < N_SEG; t += CB) base[t] = cnt[t] ? atomicAdd(&cursor[t], cnt[t]) : 0u;
    __syncthreads();
    for (int k = b0; k <= b1; k++) {
        const int sg = cls * NBANDS + k;
        const uint32_t pos = start[sg] + base[sg] + atomicAdd(&rank[sg], 1u);
        if (pos < (uint32_t)cap) ekeys[pos] = ((uint64_t)fk << 32) | (uint64_t)(uint32_t)i;
    }
}
constexpr int SEG_PARTS = 4;  // workgroups per segment: each ranks a quarter of the segment's entries against all of them
__global__ __launch_bounds__(SEG_SORT_THREADS) void k_seg_sort(ContactDev d, Bands B, const uint32_t* __restrict__ hist, const uint64_t* __restrict__ ekeys, int cap,
                                                             uint32_t* __restrict__ sidx, float* __restrict__ s_aabb, float* __restrict__ s_lo, int* __restrict__ seg,
                                                             int* __restrict__ counters)
{
    __shared__ uint64_t sh[SEG_SORT_MAX];
    __shared__ uint32_t hs[N_SEG];
    __shared__ uint32_t s_start, s_n, s_total;
    const int sg = blockIdx.x / SEG_PARTS, part = blockIdx.x % SEG_PARTS;
    for (int t = threadIdx.x; t < N_SEG; t += SEG_SORT_THREADS) hs[t] = hist[t];
    __syncthreads();
    if (threadIdx.x < 64) {  // one wavefront adds the counts before this segment and all of them
        uint32_t before = 0, all = 0;
        for (int t = threadIdx.x; t < N_SEG; t += 64) {
            all += hs[t];
            if (t < sg) before += hs[t];
        }
        for (int dd = 32; dd >= 1; dd >>= 1) {
            before += __shfl_down(before, dd, 64);
            all += __shfl_down(all, dd, 64);
        }
        if (threadIdx.x == 0) {
            s_start = before;
            s_n = hs[sg];
            s_total = all;
            if (part == 0) seg[sg] = (int)before;
            if (blockIdx.x == 0) seg[N_SEG] = all > (uint32_t)cap ? 0 : (int)all;  // (a list that did not fit is not swept: the host grows it and searches again)
        }
    }
    __syncthreads();
    const uint32_t start = s_start, n = s_n;
    if (s_total > (uint32_t)cap) return;  // (the list did not fit: the host grows it and searches again)
    if (n > (uint32_t)SEG_SORT_MAX) {
        if (threadIdx.x == 0) counters[58 - 48] = 1;  // (counters = the search's counters + 48)
        return;
    }
    if (n == 0) return;
    // Rank sort: the whole segment in LDS, every thread counts the keys below its own (all lanes read the same LDS word: a broadcast, no
    // barrier behind the load). The keys are distinct (the primitive breaks ties), so the rank is the sorted position — the order the stable
    // radix sort left. A few hundred to a few thousand entries per segment: n^2 / 4096 LDS reads per thread.
    for (uint32_t t = threadIdx.x; t < n; t += SEG_SORT_THREADS) sh[t] = ekeys[start + t];
    __syncthreads();
    const uint32_t band = (uint32_t)(sg % NBANDS);
    const uint32_t per = (n + SEG_PARTS - 1) / SEG_PARTS, t_begin = (uint32_t)part * per, t_end = min(n, t_begin + per);
    for (uint32_t t = t_begin + threadIdx.x; t < t_end; t += SEG_SORT_THREADS) {
        const uint64_t mine = sh[t];
        uint32_t r = 0;
        for (uint32_t u = 0; u < n; u++) r += sh[u] < mine ? 1u : 0u;
        const uint32_t prim = (uint32_t)(mine & 0xffffffffull);
        const float* b = d.aabb + 6 * (size_t)prim;
        const size_t j = (size_t)start + r;
        sidx[j] = prim | (band << PRIM_BITS);
#pragma unroll
        for (int c = 0; c < 6; c++) s_aabb[6 * j + c] = b[c];
        s_lo[j] = b[B.axis];
    }
}
// Binary search of the sorted lower bounds by a whole wavefront: every step probes 64 evenly spaced positions at once, so a range of n entries is
// narrowed in log64(n) dependent loads instead of log2(n) (a sweep entry spent most of its time in these chains). STRICT: count the
// entries < v (lower bound), otherwise the entries <= v (upper bound). All lanes return the same index.
template <bool STRICT, int SUB = 64>
__device__ __forceinline__ int wave_bound_f(const float* __restrict__ a, int lo, int hi, float v)
{
    // SUB lanes (a power of two) cooperate; the other groups of the wavefront run their own searches in the same instructions
    const int lane = threadIdx.x & 63, sl = lane & (SUB - 1), gb = lane & ~(SUB - 1);
    const unsigned long long gmask = SUB == 64 ? ~0ull : ((1ull << SUB) - 1ull);
    while (hi - lo > SUB) {
        const int step = (hi - lo + SUB - 1) / SUB;
        const int pos = lo + sl * step;
        bool below = false;
        if (pos < hi) {
            const float x = a[pos];
            below = STRICT ? (x < v) : (x <= v);
        }
        const int c = __popcll((__ballot(below) >> gb) & gmask);  // the predicate is monotone along the sorted array: lanes 0 .. c-1
        if (c == 0) return lo;
        const int new_hi = lo + c * step < hi ? lo + c * step : hi;
        lo = lo + (c - 1) * step + 1;
        hi = new_hi;
    }
    bool below = false;
    if (lo + sl < hi) {
        const float x = a[lo + sl];
        below = STRICT ? (x < v) : (x <= v);
    }
    return lo + __popcll((__ballot(below) >> gb) & gmask);
}
// SWEEP_SUB lanes per sorted entry (most ranges hold a few dozen candidates: a whole wavefront per entry was launch-bound, 0.4 M
// wavefronts for a 256 x 256 cloth). PROXIMITY: point entries look for triangles (lo in [lo, hi]), triangle entries for points
// (lo in (lo, hi]), edge entries for later edges. INTERSECTION (!PROXIMITY): edge entries look for triangles, triangle entries for edges.
// A wavefront scans at most SWEEP_SPLIT candidates itself: the rest of a long range (a floor triangle under a 256 x 256 cloth has 66 k
// points in its interval) becomes tasks of SWEEP_SPLIT candidates that k_sweep_tasks spreads over the chip; one wavefront walking such a
// range alone was 0.9 ms of a 1 ms detection.
#ifndef MISTARK_SWEEP_SUB
#define MISTARK_SWEEP_SUB 8
#endif
// lanes per sorted entry; measured 8 / 16 / 32: contact callbacks of configs[2] 1.63 / 1.73 / 1.90 ms per step, configs[3] equal within noise; round 6,
// A/B of 4 / 8 / 16 / 32 on one box (variant libraries): configs[2] tilted 144 / 145.5 / 139 / 133 Newton-steps/s, configs[4] 110 / 112.7 / 108.5 / 110,
// configs[3] 162 / 165 / 165 / 165: 8 (16 through round 5)
constexpr int SWEEP_SUB = MISTARK_SWEEP_SUB;
#ifndef MISTARK_SWEEP_SPLIT
#define MISTARK_SWEEP_SPLIT 512
#endif
constexpr int SWEEP_SPLIT = MISTARK_SWEEP_SPLIT;
constexpr int SWEEP_TASK_CAP = 1 << 16;
constexpr int SWEEP_TASK_WAVES = 4096;
struct SweepEntry
{
    int cls, tc, band, b_first, src, tgt_start, a1, a2;
    float lo, hi, lo1, hi1, lo2, hi2;
};
template <bool PROXIMITY>
__device__ __forceinline__ bool sweep_entry(const ContactDev& d, const Bands& B, const uint32_t* __restrict__ sidx, const float* __restrict__ s_aabb, const int* __restrict__ seg, int sp,
                                            int pt_on, int ee_on, SweepEntry& E)
{
    const uint32_t sw = sidx[sp];
    const int gi = (int)(sw & SIDX_PRIM);  // primitive (points | triangles | edges numbering)
    E.cls = gi < d.n_v ? 0 : (gi < d.n_v + d.n_t ? 1 : 2);
    if (PROXIMITY) {
        if (E.cls == 2 ? !ee_on : !pt_on) return false;
        E.tc = E.cls == 0 ? 1 : (E.cls == 1 ? 0 : 2);
    } else {
        if (E.cls == 0) return false;
        E.tc = E.cls == 2 ? 1 : 2;
    }
    const float* sb = s_aabb + 6 * (size_t)sp;
    const int ax = B.axis;
    E.a1 = (ax + 1) % 3;
    E.a2 = (ax + 2) % 3;
    E.lo = sb[ax]; E.hi = sb[3 + ax]; E.lo1 = sb[E.a1]; E.hi1 = sb[3 + E.a1]; E.lo2 = sb[E.a2]; E.hi2 = sb[3 + E.a2];
    // which band is this entry? the entries of a box are consecutive bands starting at its first one; k_bp_fill left it in the index word
    E.b_first = band_of(B, sb[B.band_axis]);
    E.band = (int)(sw >> PRIM_BITS);
    const int src_start = E.cls == 0 ? 0 : (E.cls == 1 ? d.n_v : d.n_v + d.n_t);
    E.tgt_start = E.tc == 0 ? 0 : (E.tc == 1 ? d.n_v : d.n_v + d.n_t);
    E.src = gi - src_start;
    return true;
}
// The narrow phase runs out of a per-wavefront queue: the lanes that scan candidate boxes only ENQUEUE the pairs whose boxes overlap (30 % of
// the candidates of a cloth), and whenever 64 pairs wait, all 64 lanes run the closest-feature classification and distance (or the
// edge-triangle test) on one pair each. Run in place, the double-precision narrow phase was executed with a third of the lanes on every trip
// of the scan loop: 190 of the intersection sweep's 345 us and 300 of the proximity sweep's 478 us on configs[2].
constexpr int SWEEP_QUEUE = 128;  // per wavefront: at most 63 waiting + 64 new
struct WaveQueue
{
    uint64_t* buf;  // LDS, SWEEP_QUEUE entries of this wavefront
    int count;      // wave-uniform
};
// pair = kind[63:62] (0 point-triangle, 1 edge-edge, 2 edge-triangle intersection) | a[51:26] | b[25:0]
__device__ __forceinline__ uint64_t pack_pair(int kind, int a, int b) { return ((uint64_t)kind << 62) | ((uint64_t)a << PRIM_BITS) | (uint64_t)b; }
template <bool FRICTION>
__device__ __forceinline__ int narrow_pair(const ContactDev& d, uint64_t pr, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    const int kind = (int)(pr >> 62), a = (int)((pr >> PRIM_BITS) & ((1u << PRIM_BITS) - 1)), b = (int)(pr & ((1u << PRIM_BITS) - 1));
    if (kind == 0) narrow_pt<FRICTION>(d, a, b, enl2, keys, counters, key_cap);
    else if (kind == 1) narrow_ee<FRICTION>(d, a, b, enl2, keys, counters, key_cap);
    else {
        const int e = a, t = b;
        const int v0 = d.edge[2 * e], v1 = d.edge[2 * e + 1], u0 = d.tri[3 * t], u1 = d.tri[3 * t + 1], u2 = d.tri[3 * t + 2];
        if (v0 == u0 || v0 == u1 || v0 == u2 || v1 == u0 || v1 == u1 || v1 == u2) return 0;  // BroadPhaseET.cpp:161-165
        if (d.disabled[d.edge_mesh[e] * d.n_mesh + d.tri_mesh[t]]) return 0;
        if (edge_intersects_triangle(ldx(d.X, v0), ldx(d.X, v1), ldx(d.X, u0), ldx(d.X, u1), ldx(d.X, u2))) {
            if (enl2 < 0.0) push_key(pack_key(0, 0, 0, e, t), keys, counters, key_cap);  // (enl2 < 0: the caller wants the pairs, not only their number)
            return 1;
        }
    }
    return 0;
}
// every lane of the wavefront must call this (lanes without a pair pass has = false); returns the lane's intersection hits
template <bool FRICTION>
__device__ __forceinline__ int queue_push(WaveQueue& q, bool has, uint64_t pr, const ContactDev& d, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(has);
    if (m == 0ull) return 0;
    if (has) q.buf[q.count + __popcll(m & ((1ull << lane) - 1ull))] = pr;
    q.count += __popcll(m);
    int hits = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (q.count >= 64) {
        const uint64_t mine = q.buf[lane];
        const uint64_t tail = lane + 64 < q.count ? q.buf[lane + 64] : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane + 64 < q.count) q.buf[lane] = tail;
        q.count -= 64;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        hits = narrow_pair<FRICTION>(d, mine, enl2, keys, counters, key_cap);
    }
    return hits;
}
template <bool FRICTION>
__device__ __forceinline__ int queue_flush(WaveQueue& q, const ContactDev& d, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap)
{
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int hits = 0;
    if (lane < q.count) hits = narrow_pair<FRICTION>(d, q.buf[lane], enl2, keys, counters, key_cap);
    q.count = 0;
    return hits;
}
// candidates [j0, j1) of one entry (SUB lanes share an entry; `valid` = this lane's group has one), wave-uniform loop: every lane of the
// wavefront takes part until the longest range of the wavefront is done. Returns this lane's intersection count (!PROXIMITY).
template <bool PROXIMITY, bool FRICTION, int SUB>
__device__ __forceinline__ int sweep_scan(const ContactDev& d, const Bands& B, const uint32_t* __restrict__ sidx, const float* __restrict__ s_aabb, const SweepEntry& E, bool valid, int j0,
                                          int j1, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap, WaveQueue& q)
{
    const int sl = threadIdx.x & (SUB - 1);
    int hits = 0;
    if (!valid) j1 = j0;
    // (no local arrays indexed at run time: they would live in scratch memory)
    for (int j = j0 + sl; __any(j < j1); j += SUB) {
        bool has = false;
        uint64_t pr = 0ull;
        if (j < j1) {
            const float* tb = s_aabb + 6 * (size_t)j;
            if (E.lo1 <= tb[3 + E.a1] && tb[E.a1] <= E.hi1 && E.lo2 <= tb[3 + E.a2] && tb[E.a2] <= E.hi2) {
                const int tb_first = band_of(B, tb[B.band_axis]);
                if (E.band == (E.b_first > tb_first ? E.b_first : tb_first)) {  // the pair is reported in its first common band only
                    const int tgt = (int)(sidx[j] & SIDX_PRIM) - E.tgt_start, src = E.src;
                    has = true;
                    if (PROXIMITY) {
                        if (E.cls == 0) pr = pack_pair(0, src, tgt);
                        else if (E.cls == 1) pr = pack_pair(0, tgt, src);
                        else pr = pack_pair(1, src < tgt ? src : tgt, src < tgt ? tgt : src);
                    } else {
                        // the boxes of this list are the proximity search's (enlarged by the contact thickness, so that both searches share one
                        // sort): the intersection test needs the tight ones to overlap — 4 of 5 candidates of a cloth end here, before any index
                        // or position is loaded
                        const int ax = B.axis;
                        has = E.lo + B.shrink <= tb[3 + ax] && tb[ax] + B.shrink <= E.hi && E.lo1 + B.shrink <= tb[3 + E.a1] && tb[E.a1] + B.shrink <= E.hi1 &&
                              E.lo2 + B.shrink <= tb[3 + E.a2] && tb[E.a2] + B.shrink <= E.hi2;
                        pr = E.cls == 2 ? pack_pair(2, src, tgt) : pack_pair(2, tgt, src);
                    }
                }
            }
        }
        // (the intersection test is cheap and most box pairs of an edge and a triangle survive to it: run in place it is faster, 345 against 375 us)
        if (PROXIMITY) hits += queue_push<FRICTION>(q, has, pr, d, enl2, keys, counters, key_cap);
        else if (has) hits += narrow_pair<FRICTION>(d, pr, enl2, keys, counters, key_cap);
    }
    return hits;
}
template <bool PROXIMITY, bool FRICTION>
__global__ __launch_bounds__(CB) void k_sweep(ContactDev d, Bands B, const uint32_t* __restrict__ sidx, const float* __restrict__ s_aabb, const float* __restrict__ s_lo,
                                              const int* __restrict__ seg, int pt_on, int ee_on, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap,
                                              int* __restrict__ task_count, int* __restrict__ tasks, int shard_rank, int shard_world)
{
    constexpr int SUB = SWEEP_SUB;
    __shared__ uint64_t q_buf[CB / 64][SWEEP_QUEUE];
    WaveQueue q{q_buf[threadIdx.x >> 6], 0};
    const int lane = threadIdx.x & 63, sl = lane & (SUB - 1), gb = lane & ~(SUB - 1);
    // (sharded search: the workgroups of the full grid are dealt out to the ranks round robin — neighbouring workgroups hold entries of the same
    // class and band, so every rank gets its share of points, triangles and edges)
    const int sp = (blockIdx.x * shard_world + shard_rank) * (CB / SUB) + threadIdx.x / SUB;
    SweepEntry E{};
    bool valid = sp < seg[3 * NBANDS];
    if (valid) valid = sweep_entry<PROXIMITY>(d, B, sidx, s_aabb, seg, sp, pt_on, ee_on, E);
    int j0 = 0, j_own = 0, r0 = 0, r1 = 0;  // own range [j0, j_own), remainder of a long range that found no room in the task list [r0, r1)
    if (valid) {  // (uniform within a group of SUB lanes: the searches' ballots see whole groups)
        const unsigned long long gmask = SUB == 64 ? ~0ull : ((1ull << SUB) - 1ull);
        const int t_begin = seg[E.tc * NBANDS + E.band], t_end = seg[E.tc * NBANDS + E.band + 1];
        if (E.cls == 2 && E.tc == 2) j0 = sp + 1;                                   // later edges of the same segment
        else {
            // Both segments — the entry's own and its targets' — are sorted by the lower bound along the sweep axis over the same band, so
            // the entry's relative position in its segment is where its range starts in the other one, give or take a few entries: ONE probe of
            // SUB consecutive lower bounds around that position finds the bound exactly (a[j - 1] < v <= a[j] is seen inside the window) for
            // almost every entry of a mesh; the SUB-ary search (three dependent loads on a 3 k-entry segment) remains for the others.
            const bool strict = E.cls == 0 || (!PROXIMITY && E.cls == 2);  // target lo in [lo, hi] / in (lo, hi]: the other direction took ties
            const int o_begin = seg[E.cls * NBANDS + E.band], o_end = seg[E.cls * NBANDS + E.band + 1];
            const int n_own = o_end - o_begin > 1 ? o_end - o_begin : 1;
            int g = t_begin + (int)((long long)(sp - o_begin) * (long long)(t_end - t_begin) / n_own) - SUB / 2;
            if (g > t_end - SUB) g = t_end - SUB;
            if (g < t_begin) g = t_begin;
            bool below = false;
            if (g + sl < t_end) {
                const float x = s_lo[g + sl];
                below = strict ? (x < E.lo) : (x <= E.lo);
            }
            const int cb = __popcll((__ballot(below) >> gb) & gmask);
            if ((cb > 0 || g == t_begin) && (cb < SUB || g + SUB >= t_end)) j0 = g + cb;
            else if (strict) j0 = wave_bound_f<true, SUB>(s_lo, t_begin, t_end, E.lo);
            else j0 = wave_bound_f<false, SUB>(s_lo, t_begin, t_end, E.lo);
        }
        // the end of the range: most ranges are shorter than SUB candidates — one probe of the SUB lower bounds behind j0 then holds it
        int j1;
        {
            const int jb = j0 > t_begin ? j0 : t_begin;
            bool inside = false;
            if (jb + sl < t_end) inside = s_lo[jb + sl] <= E.hi;
            const int ci = __popcll((__ballot(inside) >> gb) & gmask);
            if (ci < SUB || jb + SUB >= t_end) j1 = jb + ci;
            else j1 = wave_bound_f<false, SUB>(s_lo, jb + SUB, t_end, E.hi);
        }
        j_own = j1;
        if (j1 - j0 > SWEEP_SPLIT) {  // long range: hand the rest out in tasks (those that fit the list; the remainder stays here)
            const int first = j0 + SWEEP_SPLIT;
            const int n_task = (j1 - first + SWEEP_SPLIT - 1) / SWEEP_SPLIT;
            int base = 0;
            if (sl == 0) base = atomicAdd(task_count, n_task);
            base = __shfl(base, gb, 64);
            const int n_fit = base >= SWEEP_TASK_CAP ? 0 : (n_task < SWEEP_TASK_CAP - base ? n_task : SWEEP_TASK_CAP - base);
            for (int t = sl; t < n_fit; t += SUB) {
                int* T = tasks + 3 * (size_t)(base + t);
                T[0] = sp;
                T[1] = first + t * SWEEP_SPLIT;
                T[2] = first + (t + 1) * SWEEP_SPLIT < j1 ? first + (t + 1) * SWEEP_SPLIT : j1;
            }
            j_own = first;
            if (n_fit < n_task) {  // (list full)
                r0 = first + n_fit * SWEEP_SPLIT;
                r1 = j1;
            }
        }
    }
    int hits = sweep_scan<PROXIMITY, FRICTION, SUB>(d, B, sidx, s_aabb, E, valid, j0, j_own, enl2, keys, counters, key_cap, q);
    if (__any(r1 > r0)) hits += sweep_scan<PROXIMITY, FRICTION, SUB>(d, B, sidx, s_aabb, E, valid, r0, r1, enl2, keys, counters, key_cap, q);
    hits += queue_flush<FRICTION>(q, d, enl2, keys, counters, key_cap);
    if (!PROXIMITY && hits) atomicAdd(&counters[1], hits);
}
template <bool PROXIMITY, bool FRICTION>
__global__ __launch_bounds__(CB) void k_sweep_tasks(ContactDev d, Bands B, const uint32_t* __restrict__ sidx, const float* __restrict__ s_aabb, const int* __restrict__ seg, int pt_on,
                                                    int ee_on, double enl2, uint64_t* __restrict__ keys, int* __restrict__ counters, int key_cap,
                                                    const int* __restrict__ task_count, const int* __restrict__ tasks)
{
    __shared__ uint64_t q_buf[CB / 64][SWEEP_QUEUE];
    WaveQueue q{q_buf[threadIdx.x >> 6], 0};
    const int w = blockIdx.x * (CB / 64) + (threadIdx.x >> 6);
    const int n = task_count[0] < SWEEP_TASK_CAP ? task_count[0] : SWEEP_TASK_CAP;
    int hits = 0;
    for (int t = w; t < n; t += SWEEP_TASK_WAVES) {  // (wave-uniform: one task per wavefront and trip)
        const int* T = tasks + 3 * (size_t)t;
        SweepEntry E{};
        const bool valid = sweep_entry<PROXIMITY>(d, B, sidx, s_aabb, seg, T[0], pt_on, ee_on, E);
        hits += sweep_scan<PROXIMITY, FRICTION, 64>(d, B, sidx, s_aabb, E, valid, T[1], T[2], enl2, keys, counters, key_cap, q);
    }
    hits += queue_flush<FRICTION>(q, d, enl2, keys, counters, key_cap);
    if (!PROXIMITY && hits) atomicAdd(&counters[1], hits);
}

// ---- sorted keys -> tables -----------------------------------------------------------------------------------------------------------
// bounds[t] = first sorted key of table t (t = 0..N_TABLES), same[0] = 1 iff the list equals the previous one
// keys[n .. n_pad) = padding that sorts last (n = the count the search left on the device)
__global__ __launch_bounds__(CB) void k_pad_keys(uint64_t* __restrict__ keys, const int* __restrict__ n_dev, int n_pad)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    if (i >= *n_dev && i < n_pad) keys[i] = ~0ull;
}
// (n_dev: the number of keys as the search left it on the device, capped by n — the list was sorted over a padded length before the host
// knew the count)
__global__ __launch_bounds__(CB) void k_table_bounds(const uint64_t* __restrict__ keys, int n, const int* __restrict__ n_dev, const uint64_t* __restrict__ prev, int n_prev,
                                                    int* __restrict__ bounds, int* __restrict__ differs)
{
    if (n_dev) n = min(n, *n_dev);
    const int i = blockIdx.x * CB + threadIdx.x;
    if (i > n) return;
    const int t_here = i < n ? (int)(keys[i] >> 58) : N_TABLES;
    const int t_prev = i > 0 ? (int)(keys[i - 1] >> 58) : -1;
    for (int t = t_prev + 1; t <= t_here; t++) bounds[t] = i;
    if (i < n && (n != n_prev || keys[i] != prev[i])) differs[0] = 1;
}
struct Side
{
    int mesh, nv, has_edge;
    int cv[3], ecv[2];
};
__device__ __forceinline__ void decode(const ContactDev& d, uint64_t key, int& fam, Side& A, Side& B)
{
    const int src = (int)((key >> 57) & 1), type = (int)((key >> 53) & 15);
    const int a = (int)((key >> PRIM_BITS) & ((1u << PRIM_BITS) - 1)), b = (int)(key & ((1u << PRIM_BITS) - 1));
    if (src == 0) {
        fam = pt_family(type);
        A.mesh = d.cv_mesh[a]; A.nv = 1; A.cv[0] = a; A.has_edge = 0;
        B.mesh = d.tri_mesh[b]; B.has_edge = 0;
        const int* t = d.tri + 3 * b;
        if (fam == 0) {
            B.nv = 1;
            B.cv[0] = t[type];
        } else if (fam == 1) {  // P_E0: (t0,t1), P_E1: (t1,t2), P_E2: (t2,t0) (ProximityDetection.cpp:117-119)
            const int k = type - P_E0;
            B.nv = 2;
            B.cv[0] = t[k];
            B.cv[1] = t[(k + 1) % 3];
        } else {
            B.nv = 3;
            B.cv[0] = t[0]; B.cv[1] = t[1]; B.cv[2] = t[2];
        }
        return;
    }
    fam = ee_family(type);
    const int ea[2] = {d.edge[2 * a], d.edge[2 * a + 1]}, eb[2] = {d.edge[2 * b], d.edge[2 * b + 1]};
    const int ma = d.edge_mesh[a], mb = d.edge_mesh[b];
    auto edge_side = [](Side& S, int mesh, const int* e) {
        S.mesh = mesh; S.nv = 2; S.cv[0] = e[0]; S.cv[1] = e[1]; S.has_edge = 1; S.ecv[0] = e[0]; S.ecv[1] = e[1];
    };
    auto point_side = [](Side& S, int mesh, const int* e, int v) {
        S.mesh = mesh; S.nv = 1; S.cv[0] = v; S.has_edge = 1; S.ecv[0] = e[0]; S.ecv[1] = e[1];
    };
    switch (type) {  // ProximityDetection.cpp:166-176
        case EA0_EB0: point_side(A, ma, ea, ea[0]); point_side(B, mb, eb, eb[0]); break;
        case EA0_EB1: point_side(A, ma, ea, ea[0]); point_side(B, mb, eb, eb[1]); break;
        case EA1_EB0: point_side(A, ma, ea, ea[1]); point_side(B, mb, eb, eb[0]); break;
        case EA1_EB1: point_side(A, ma, ea, ea[1]); point_side(B, mb, eb, eb[1]); break;
        case EA_EB0: point_side(A, mb, eb, eb[0]); edge_side(B, ma, ea); break;
        case EA_EB1: point_side(A, mb, eb, eb[1]); edge_side(B, ma, ea); break;
        case EA0_EB: point_side(A, ma, ea, ea[0]); edge_side(B, mb, eb); break;
        case EA1_EB: point_side(A, ma, ea, ea[1]); edge_side(B, mb, eb); break;
        default: edge_side(A, ma, ea); edge_side(B, mb, eb); break;
    }
}
// columns one side contributes to a BARRIER row: point-triangle families list the vertices, edge-edge families the edge then
// the point (contact_and_friction_data.h)
__device__ __forceinline__ int barrier_cols(const ContactDev& d, const Side& S, int fam, int* out)
{
    int n = 0;
    if (fam >= 3) {
        out[n++] = d.cv_src[S.ecv[0]];
        out[n++] = d.cv_src[S.ecv[1]];
        if (S.nv == 1) out[n++] = d.cv_src[S.cv[0]];
    } else {
        for (int i = 0; i < S.nv; i++) out[n++] = d.cv_src[S.cv[i]];
    }
    return n;
}
__global__ __launch_bounds__(CB) void k_route(ContactDev d, const uint64_t* __restrict__ keys, int n, const TableDev* __restrict__ tables, const double* __restrict__ kptr)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const int table = (int)(key >> 58);
    const TableDev T = tables[table];
    const int row = i - T.start;
    int fam;
    Side A, B;
    decode(d, key, fam, A, B);
    const int ka = d.mesh_kind[A.mesh], kb = d.mesh_kind[B.mesh];
    const int ia = d.mesh_idx[A.mesh], ib = d.mesh_idx[B.mesh];
    int32_t* out = T.conn + (size_t)row * T.stride;
    int n_out = 0;
    if (table < N_CONTACT_TABLES) {
        int ca[3], cb[3];
        const int na = barrier_cols(d, A, fam, ca), nb = barrier_cols(d, B, fam, cb);
        if (ka == 0 && kb == 1) {  // deformable first: {B.group, A.group, B.idx_in_ps, B..., A...} (EnergyFrictionalContact.cpp:398-399 etc.)
            out[n_out++] = B.mesh; out[n_out++] = A.mesh; out[n_out++] = ib;
            for (int k = 0; k < nb; k++) out[n_out++] = cb[k];
            for (int k = 0; k < na; k++) out[n_out++] = ca[k];
        } else {
            out[n_out++] = A.mesh; out[n_out++] = B.mesh;
            if (ka == 1) out[n_out++] = ia;
            if (ka == 1 && kb == 1) out[n_out++] = ib;
            for (int k = 0; k < na; k++) out[n_out++] = ca[k];
            for (int k = 0; k < nb; k++) out[n_out++] = cb[k];
        }
        return;
    }
    // friction row: running index, rigid body indices, vertices (:592-772)
    out[n_out++] = row;
    if (ka == 0 && kb == 1) {
        out[n_out++] = ib;
        for (int k = 0; k < B.nv; k++) out[n_out++] = d.cv_src[B.cv[k]];
        for (int k = 0; k < A.nv; k++) out[n_out++] = d.cv_src[A.cv[k]];
    } else {
        if (ka == 1) out[n_out++] = ia;
        if (ka == 1 && kb == 1) out[n_out++] = ib;
        for (int k = 0; k < A.nv; k++) out[n_out++] = d.cv_src[A.cv[k]];
        for (int k = 0; k < B.nv; k++) out[n_out++] = d.cv_src[B.cv[k]];
    }
    // contact data in the detected (A, B) order, whatever the row order (the lambdas of :548-587)
    D3 xa[2], xb[3];
    for (int k = 0; k < A.nv && k < 2; k++) xa[k] = ldx(d.X, A.cv[k]);
    for (int k = 0; k < B.nv; k++) xb[k] = ldx(d.X, B.cv[k]);
    double dist;
    double* Tm = T.T + 6 * (size_t)row;
    double* bary = T.bary ? T.bary + (size_t)T.nbary * row : nullptr;
    if (fam == 0 || fam == 3) {
        basis_point_point(xa[0], xb[0], Tm);
        dist = ::sqrt(sq3(xb[0] - xa[0]));
    } else if (fam == 1 || fam == 4) {
        bary_point_edge(xa[0], xb[0], xb[1], bary);
        basis_point_edge(xa[0], xb[0], xb[1], Tm);
        dist = ::sqrt(point_line_sq(xa[0], xb[0], xb[1]));
    } else if (fam == 2) {
        bary_point_triangle(xa[0], xb[0], xb[1], xb[2], bary);
        basis_triangle(xb[0], xb[1], xb[2], Tm);
        const D3 nrm = cross3(xb[1] - xb[0], xb[2] - xb[0]);
        const double h = dot3(xa[0] - xb[0], nrm);
        dist = ::sqrt(h * h / sq3(nrm));
    } else {
        bary_edge_edge(xa[0], xa[1], xb[0], xb[1], bary);
        basis_edge_edge(xa[0], xa[1], xb[0], xb[1], Tm);
        const D3 nrm = cross3(xa[1] - xa[0], xb[1] - xb[0]);
        const double h = dot3(xb[0] - xa[0], nrm);
        dist = ::sqrt(h * h / sq3(nrm));
    }
    const double dhat = d.thick[A.mesh] + d.thick[B.mesh];
    T.mu[row] = d.mu[A.mesh * d.n_mesh + B.mesh];
    T.fn[row] = kptr[0] * (dhat - dist) * (dhat - dist);  // _barrier_force, cubic barrier (:1238-1242)
}
}  // namespace

// ======================================================================================================================================
struct ContactSystem
{
    mistark_contact_arrays arr{};
    struct Mesh
    {
        int kind, idx_in_ps, v_off, n_v, t_off, n_t, e_off, n_e;
    };
    std::vector<Mesh> meshes;
    std::vector<int32_t> h_cv_src, h_cv_mesh, h_tri, h_tri_mesh, h_edge, h_edge_mesh;
    std::map<std::pair<int, int>, double> friction;
    std::vector<std::pair<int, int>> disabled_pairs;
    std::vector<int32_t> h_bl_pt, h_bl_ee;  // range blacklists, 4 ints each (ContactDev::bl_pt / bl_ee)
    DevBuf<int32_t> bl_pt, bl_ee;
    bool meshes_dirty = true;
    bool pt_enabled = true, ee_enabled = true;
    // sharded search (merge_sharded_search): keys per rank in the exchange, its buffers, how many searches took it
    int shard_k = 0;
    DevBuf<double> shard_send, shard_recv;
    int64_t n_sharded_searches = 0;
    // barrier-table searches that ran / that ran at the very state (Context::data_version, dt) the previous one had searched (counter
    // "contact_repeated_searches": 0 since the cache is refreshed by a search that finds the installed tables unchanged)
    int64_t n_searches = 0, n_repeated_searches = 0;
    uint64_t searched_version = 0;
    double searched_dt = -1.0;

    DevBuf<int32_t> cv_src, cv_mesh, tri, tri_mesh, edge, edge_mesh, mesh_kind, mesh_idx;
    DevBuf<uint8_t> disabled;
    DevBuf<double> mu, X;
    DevBuf<float> aabb;
    DevBuf<uint64_t> keys, keys_alt, prev;
    // sweep and prune
    DevBuf<uint64_t> bp_keys, bp_keys_alt;
    DevBuf<uint32_t> bp_idx, bp_idx_alt;
    DevBuf<float> s_aabb, s_lo;
    DevBuf<uint32_t> bp_cnt, bp_off;
    DevBuf<uint32_t> bp_hist;   // entries per (class, band) segment | cursors (k_bp_hist / k_bp_scatter)
    bool seg_sort_ok = true;    // false once a segment exceeded SEG_SORT_MAX entries (counters[58]): the library sort from then on
    DevBuf<int> seg;
    const uint32_t* s_idx = nullptr;
    Bands bands{-1, -1, 0.f, 1.f, 0.f};
    int bp_cap = 0;
    int64_t n_updates = 0;
    bool brute_force = false;  // ablation / fallback: LDS-tiled all-pairs kernels
    int64_t n_prev = -1;  // keys of the barrier tables currently installed (-1: none)
    DevBuf<int> counters;  // [0] candidates, [1] intersections, [2] differs, [8..8+N_TABLES] bounds, [48..51] box list (k_bp_fill; [51] = entries needed)
    // result of the last barrier-table search and the inputs it saw (Context::data_version, dt): an identical request is answered from here
    int n_last = 0;  // keys found by the last search (sizes the padded sort of the next one)
    // The intersection check of a line-search candidate runs the proximity search of the evaluation that follows it, too (same state, same
    // boxes; one read-back for both): the result waits here until detect_and_route asks for exactly that state.
    struct Speculated
    {
        bool valid = false;
        uint64_t version = 0;
        double dt = 0.0;
        float enl = -1.f;
        int n = 0;
        int h[64];
        const uint64_t* sorted = nullptr;
    } spec;
    bool cache_valid = false;
    bool ix_valid = false;  // last intersection count (count_intersections)
    uint64_t ix_version = 0;
    double ix_dt = 0.0;
    int64_t ix_n = 0;
    // the sorted box list of the last search: reused when the next search sees the same state with the same enlargement (the intersection
    // check of a line-search candidate and the proximity search of the energy evaluation that follows it)
    bool bp_valid = false;
    uint64_t bp_version = 0;
    double bp_dt = 0.0;
    float bp_enl = -1.f;
    uint64_t cache_version = 0;
    double cache_dt = 0.0;
    int64_t cache_n = 0;
    DevBuf<double> thick_override;  // standalone detector (mistark_cd_*): no thickness filter, one huge value per mesh
    DevBuf<int> sweep_tasks;  // (entry, first candidate, end) of the split-off parts of long sweep ranges + their count
    DevBuf<uint8_t> cub_tmp;
    DevBuf<TableDev> tables_dev;
    size_t key_cap = 0;
    TableDev* td_pinned = nullptr;

    struct Table
    {
        int pot = -1, stride = 0, nbary = 0, n = 0;
        int a_T = -1, a_mu = -1, a_fn = -1, a_bary = -1;
        DevBuf<int32_t> conn;
    };
    Table tables[N_TABLES];
    int n_v = 0, n_t = 0, n_e = 0;
};
// sharded runs: the block rows contact potentials may reference (the collision vertices of the deformable meshes; rigid bodies are small DoF
// sets, shared anyway). Every rank keeps them as ghosts (shard.hip).
int64_t contact_sharded_searches(const Context& c) { return c.contact ? c.contact->n_sharded_searches : 0; }
int64_t contact_searches(const Context& c, bool repeated) { return !c.contact ? 0 : (repeated ? c.contact->n_repeated_searches : c.contact->n_searches); }
void contact_shared_rows(Context& c, std::vector<int32_t>& rows)
{
    if (!c.contact) return;
    ContactSystem& cs = *c.contact;
    const int v1 = cs.arr.v1;
    if (v1 < 0 || v1 >= (int)c.arrays.size() || c.arrays[v1].dof_set < 0) return;
    const int64_t row0 = c.dof_sets[c.arrays[v1].dof_set].offset / 3;
    for (const ContactSystem::Mesh& m : cs.meshes)
        if (m.kind == MISTARK_CONTACT_DEFORMABLE)
            for (int i = 0; i < m.n_v; i++) rows.push_back((int32_t)(row0 + cs.h_cv_src[(size_t)(m.v_off + i)]));
}
void contact_destroy(ContactSystem* cs)
{
    if (cs && cs->td_pinned) (void)hipHostFree(cs->td_pinned);
    delete cs;
}

namespace {
ContactSystem& CS(Context& c)
{
    if (!c.contact) throw Error("contact: call mistark_contact_init first");
    return *c.contact;
}
int new_device_array(Context& c, int stride)
{
    c.arrays.emplace_back();
    Array& a = c.arrays.back();
    a.host = nullptr;  // device-only: filled by k_route
    a.n_items = 0;
    a.stride = stride;
    a.need_upload = false;
    c.layout_dirty = true;
    return (int)c.arrays.size() - 1;
}
void contact_init(Context& c, const mistark_contact_arrays& arr)
{
    if (c.contact) throw Error("contact: already initialised");
    auto* cs = new ContactSystem();
    c.contact = cs;
    cs->arr = arr;
    const int32_t* ids = &arr.v1;
    const bool has_d = arr.v1 >= 0 && arr.x0 >= 0 && arr.X >= 0;
    const bool has_rb = arr.rb_xloc >= 0 && arr.rb_v1 >= 0 && arr.rb_w1 >= 0 && arr.rb_t0 >= 0 && arr.rb_q0 >= 0;
    if (arr.dt < 0 || arr.k < 0 || arr.thickness < 0 || arr.epsv < 0) throw Error("contact: dt, k, thickness and epsv arrays are required");
    for (int t = 0; t < N_TABLES; t++) {
        ContactSystem::Table& T = cs->tables[t];
        std::vector<Bind> rec = recipe(t, T.stride);
        T.nbary = table_nbary(t);
        bool usable = true;
        for (const Bind& b : rec) {
            if (b.role <= R_X && !has_d) usable = false;
            if (b.role >= R_RB_XLOC && b.role <= R_RB_Q0 && !has_rb) usable = false;
        }
        if (!usable) continue;  // the scene has no such physical system: the table can never receive a row
        if (t >= N_CONTACT_TABLES) {
            T.a_T = new_device_array(c, 6);
            T.a_mu = new_device_array(c, 1);
            T.a_fn = new_device_array(c, 1);
            if (T.nbary) T.a_bary = new_device_array(c, T.nbary);
        }
        std::vector<mistark_binding> bs;
        for (const Bind& b : rec) {
            int id;
            switch (b.role) {
                case R_T: id = T.a_T; break;
                case R_MU: id = T.a_mu; break;
                case R_FN: id = T.a_fn; break;
                case R_BARY: id = T.a_bary; break;
                default: id = ids[b.role];
            }
            bs.push_back(mistark_binding{id, b.stride, b.col});
        }
        T.pot = register_potential(c, TABLE_NAMES[t], nullptr, 0, T.stride, bs.data(), (int)bs.size());
        c.pots[T.pot].part = 1;  // contact tables change inside the Newton loop: dynamic matrix part
        T.conn.ensure(64 * (size_t)T.stride);
        c.pots[T.pot].conn_ext = T.conn.p;
        c.pots[T.pot].conn_dirty = false;
    }
    c.layout_dirty = true;
    c.part[0].dirty = c.part[1].dirty = true;
    cs->counters.ensure(64);
    cs->tables_dev.ensure(N_TABLES);
}
int contact_add_mesh(Context& c, int kind, int idx_in_ps, const int32_t* vidx, int nv, const int32_t* tris, int nt, const int32_t* edges, int ne)
{
    ContactSystem& cs = CS(c);
    if (kind != MISTARK_CONTACT_DEFORMABLE && kind != MISTARK_CONTACT_RIGIDBODY) throw Error("contact: bad mesh kind");
    if (nv <= 0 || nt < 0 || ne < 0) throw Error("contact: bad mesh size");
    ContactSystem::Mesh m{kind, idx_in_ps, cs.n_v, nv, cs.n_t, nt, cs.n_e, ne};
    const int g = (int)cs.meshes.size();
    c.layout_dirty = true;  // (sharded runs: the mesh's vertices become shared rows)
    for (int i = 0; i < nv; i++) {
        cs.h_cv_src.push_back(vidx[i]);
        cs.h_cv_mesh.push_back(g);
    }
    for (int i = 0; i < 3 * nt; i++) {
        if (tris[i] < 0 || tris[i] >= nv) throw Error("contact: triangle vertex out of range");
        cs.h_tri.push_back(cs.n_v + tris[i]);
    }
    for (int i = 0; i < nt; i++) cs.h_tri_mesh.push_back(g);
    for (int i = 0; i < 2 * ne; i++) {
        if (edges[i] < 0 || edges[i] >= nv) throw Error("contact: edge vertex out of range");
        cs.h_edge.push_back(cs.n_v + edges[i]);
    }
    for (int i = 0; i < ne; i++) cs.h_edge_mesh.push_back(g);
    cs.n_v += nv;
    cs.n_t += nt;
    cs.n_e += ne;
    // (a sorted box entry packs the GLOBAL primitive index — vertices, then triangles, then edges — into PRIM_BITS bits beside its band: the
    // sum has to fit, not only each class)
    if ((long long)cs.n_v + (long long)cs.n_t + (long long)cs.n_e >= (1ll << PRIM_BITS))
        throw Error("contact: too many collision primitives (vertices + triangles + edges must stay below 2^" + std::to_string(PRIM_BITS) + ")");
    cs.meshes.push_back(m);
    if (kind == MISTARK_CONTACT_RIGIDBODY) cs.disabled_pairs.push_back({g, g});
    cs.meshes_dirty = true;
    cs.n_prev = -1;
    return g;
}
template <class T>
void upload(Context& c, DevBuf<T>& dst, const std::vector<T>& src)
{
    dst.ensure(std::max<size_t>(src.size(), 1));
    if (!src.empty()) MS_CHECK(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c.stream));
}
void upload_meshes(Context& c, ContactSystem& cs)
{
    if (!cs.meshes_dirty) return;
    cs.bp_valid = false;
    const int nm = (int)cs.meshes.size();
    std::vector<int32_t> kind(nm), idx(nm);
    for (int g = 0; g < nm; g++) {
        kind[g] = cs.meshes[g].kind;
        idx[g] = cs.meshes[g].idx_in_ps;
    }
    std::vector<uint8_t> dis((size_t)nm * nm, 0);
    for (auto& p : cs.disabled_pairs) dis[(size_t)p.first * nm + p.second] = dis[(size_t)p.second * nm + p.first] = 1;
    std::vector<double> mu((size_t)nm * nm, 0.0);
    for (auto& kv : cs.friction) mu[(size_t)kv.first.first * nm + kv.first.second] = mu[(size_t)kv.first.second * nm + kv.first.first] = kv.second;
    upload(c, cs.cv_src, cs.h_cv_src);
    upload(c, cs.cv_mesh, cs.h_cv_mesh);
    upload(c, cs.tri, cs.h_tri);
    upload(c, cs.tri_mesh, cs.h_tri_mesh);
    upload(c, cs.edge, cs.h_edge);
    upload(c, cs.edge_mesh, cs.h_edge_mesh);
    upload(c, cs.mesh_kind, kind);
    upload(c, cs.mesh_idx, idx);
    upload(c, cs.disabled, dis);
    upload(c, cs.mu, mu);
    if (!cs.h_bl_pt.empty()) upload(c, cs.bl_pt, cs.h_bl_pt);
    if (!cs.h_bl_ee.empty()) upload(c, cs.bl_ee, cs.h_bl_ee);
    cs.X.ensure(3 * (size_t)cs.n_v);
    cs.aabb.ensure(6 * (size_t)(cs.n_v + cs.n_t + cs.n_e));
    MS_CHECK(hipStreamSynchronize(c.stream));  // (the staging vectors above are temporaries)
    cs.meshes_dirty = false;
}
const double* arr_dev(Context& c, int id) { return id >= 0 ? c.arrays[id].dev : nullptr; }
ContactDev dev_view(Context& c, ContactSystem& cs)
{
    ContactDev d{};
    d.cv_src = cs.cv_src.p; d.cv_mesh = cs.cv_mesh.p; d.tri = cs.tri.p; d.tri_mesh = cs.tri_mesh.p; d.edge = cs.edge.p; d.edge_mesh = cs.edge_mesh.p;
    d.mesh_kind = cs.mesh_kind.p; d.mesh_idx = cs.mesh_idx.p; d.disabled = cs.disabled.p; d.mu = cs.mu.p;
    d.thick = cs.thick_override.p ? cs.thick_override.p : arr_dev(c, cs.arr.thickness);
    d.X = cs.X.p; d.aabb = cs.aabb.p;
    d.n_mesh = (int)cs.meshes.size(); d.n_v = cs.n_v; d.n_t = cs.n_t; d.n_e = cs.n_e;
    d.bl_pt = cs.bl_pt.p; d.bl_ee = cs.bl_ee.p;
    d.n_bl_pt = (int)(cs.h_bl_pt.size() / 4); d.n_bl_ee = (int)(cs.h_bl_ee.size() / 4);
    return d;
}
double max_thickness(Context& c, ContactSystem& cs)
{
    const Array& a = c.arrays[cs.arr.thickness];
    if (a.n_items < (int64_t)cs.meshes.size() || !a.host) throw Error("contact: the thickness array must hold one value per collision mesh");
    double m = 0.0;
    for (size_t g = 0; g < cs.meshes.size(); g++) m = std::max(m, a.host[g]);
    if (!(m > 0.0)) throw Error("contact: contact thickness must be positive");
    return m;
}
int chunk_for(int n_rows, int n_cols)
{
    // enough workgroups to fill 256 CUs several times over without making the column chunks shorter than 2 LDS tiles
    const int row_tiles = std::max(1, (n_rows + CB - 1) / CB);
    const int want_chunks = std::max(1, 2048 / row_tiles);
    int chunk = (n_cols + want_chunks - 1) / want_chunks;
    chunk = std::max(2 * CB, ((chunk + CB - 1) / CB) * CB);
    return chunk;
}
void update_vertices(Context& c, ContactSystem& cs, const ContactDev& d, double dt, float enl)
{
    hipLaunchKernelGGL(k_contact_vertices, dim3((cs.n_v + CB - 1) / CB), dim3(CB), 0, c.stream, d, arr_dev(c, cs.arr.x0), arr_dev(c, cs.arr.v1), arr_dev(c, cs.arr.rb_xloc),
                       arr_dev(c, cs.arr.rb_v1), arr_dev(c, cs.arr.rb_w1), arr_dev(c, cs.arr.rb_t0), arr_dev(c, cs.arr.rb_q0), dt, cs.X.p);
    const int np = cs.n_v + cs.n_t + cs.n_e;
    hipLaunchKernelGGL(k_contact_aabbs, dim3((np + CB - 1) / CB), dim3(CB), 0, c.stream, d, enl, cs.aabb.p);
}
void choose_axes(Context& c, ContactSystem& cs)
{
    // sweep axis = the axis along which the collision vertices are spread widest, band axis = the second; re-evaluated now and then (a stale
    // choice only costs speed: band indices are clamped, monotone functions of the coordinate). "Spread" is the VARIANCE of the vertices, not the
    // extent of their bounding box (round 6): a 256 x 256 cloth over a 2 m cube has its largest extent along z — the cube's eight vertices —
    // while 66 k of its 66 k + 8 vertices lie within 5 cm of one z; swept along z every cloth primitive met a thousand candidates per band
    // (2.3-2.8 ms per sweep, 74 % of that scene's kernel time; along x: what the flat floor box of round 4's configs[2] took, 0.3-0.4 ms).
    // The bands still cover the whole extent of their axis.
    if (cs.bands.axis >= 0 && (cs.n_updates % 256) != 0) return;
    std::vector<double> X(3 * (size_t)cs.n_v);
    MS_CHECK(hipMemcpyAsync(X.data(), cs.X.p, X.size() * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (int i = 0; i < cs.n_v; i++)
        for (int k = 0; k < 3; k++) {
            const double v = X[3 * (size_t)i + k];
            lo[k] = std::min(lo[k], v);
            hi[k] = std::max(hi[k], v);
            sum[k] += v;
            sq[k] += v * v;
        }
    double var[3];
    for (int k = 0; k < 3; k++) {
        const double m = sum[k] / std::max(cs.n_v, 1);
        var[k] = std::max(sq[k] / std::max(cs.n_v, 1) - m * m, 0.0);
        if (!(var[k] == var[k])) var[k] = 0.0;
    }
    int order[3] = {0, 1, 2};
    if (c.sweep_axis_by_extent) std::sort(order, order + 3, [&](int a, int b) { return hi[a] - lo[a] > hi[b] - lo[b]; });
    else std::stable_sort(order, order + 3, [&](int a, int b) { return var[a] > var[b]; });
    cs.bands.axis = order[0];
    cs.bands.band_axis = order[1];
    const double ext = std::max(hi[order[1]] - lo[order[1]], 1e-12);
    cs.bands.band_lo = (float)lo[order[1]];
    cs.bands.band_scale = (float)(NBANDS / ext);
}
// false: the entry list did not fit (capacity grown, caller repeats)
void sort_boxes(Context& c, ContactSystem& cs, const ContactDev& d)
{
    choose_axes(c, cs);
    cs.n_updates++;
    const int n = cs.n_v + cs.n_t + cs.n_e;
    if (cs.bp_cap < n + n / 2 + 4096) cs.bp_cap = n + n / 2 + 4096;
    const int cap = cs.bp_cap;
    cs.bp_cnt.ensure((size_t)n + 1); cs.bp_off.ensure((size_t)n + 1);
    cs.bp_keys.ensure(cap); cs.bp_keys_alt.ensure(cap); cs.bp_idx.ensure(cap); cs.bp_idx_alt.ensure(cap);
    cs.s_aabb.ensure(6 * (size_t)cap); cs.s_lo.ensure(cap); cs.seg.ensure(3 * NBANDS + 2);
    if (cs.seg_sort_ok && c.seg_sort) {  // option: three launches instead of the library's sort and the scans around it (see k_seg_sort; measured slower)
        cs.bp_hist.ensure(2 * (size_t)N_SEG);
        fill_async(c.stream, cs.bp_hist.p, 0, 2 * (size_t)N_SEG * sizeof(uint32_t));
        hipLaunchKernelGGL(k_bp_hist, dim3((n + CB - 1) / CB), dim3(CB), 0, c.stream, d, cs.bands, cs.bp_hist.p);
        hipLaunchKernelGGL(k_bp_scatter, dim3((n + CB - 1) / CB), dim3(CB), 0, c.stream, d, cs.bands, (const uint32_t*)cs.bp_hist.p, cs.bp_hist.p + N_SEG, cs.bp_keys.p, cap,
                           cs.counters.p + 48);
        hipLaunchKernelGGL(k_seg_sort, dim3(N_SEG * SEG_PARTS), dim3(SEG_SORT_THREADS), 0, c.stream, d, cs.bands, (const uint32_t*)cs.bp_hist.p, (const uint64_t*)cs.bp_keys.p, cap, cs.bp_idx.p,
                           cs.s_aabb.p, cs.s_lo.p, cs.seg.p, cs.counters.p + 48);
        cs.s_idx = cs.bp_idx.p;
        return;
    }
    hipLaunchKernelGGL(k_bp_count, dim3((n + 1 + CB - 1) / CB), dim3(CB), 0, c.stream, d, cs.bands, cs.bp_cnt.p);
    size_t tmp = 0;
    MS_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, cs.bp_cnt.p, cs.bp_off.p, n + 1, c.stream));
    cs.cub_tmp.ensure(tmp);
    MS_CHECK(hipcub::DeviceScan::ExclusiveSum(cs.cub_tmp.p, tmp, cs.bp_cnt.p, cs.bp_off.p, n + 1, c.stream));
    hipLaunchKernelGGL(k_bp_fill, dim3((n + CB - 1) / CB), dim3(CB), 0, c.stream, d, cs.bands, (const uint32_t*)cs.bp_off.p, cs.bp_keys.p, cs.bp_idx.p, cap, cs.counters.p + 48);
    hipcub::DoubleBuffer<uint64_t> dk(cs.bp_keys.p, cs.bp_keys_alt.p);
    hipcub::DoubleBuffer<uint32_t> dv(cs.bp_idx.p, cs.bp_idx_alt.p);
    MS_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, dk, dv, cap, 0, 32 + BAND_KEY_BITS, c.stream));
    cs.cub_tmp.ensure(tmp);
    MS_CHECK(hipcub::DeviceRadixSort::SortPairs(cs.cub_tmp.p, tmp, dk, dv, cap, 0, 32 + BAND_KEY_BITS, c.stream));
    cs.s_idx = dv.Current();
    hipLaunchKernelGGL(k_bp_gather, dim3((cap + 1 + CB - 1) / CB), dim3(CB), 0, c.stream, d, cs.bands, (const uint64_t*)dk.Current(), cs.s_idx, cap, cs.s_aabb.p, cs.s_lo.p, cs.seg.p);
}
// first capacity of the contact key list (it grows on demand; MISTARK_CONTACT_KEY_CAP: a small value makes the tests walk the growth path)
size_t initial_key_cap()
{
    if (const char* e = std::getenv("MISTARK_CONTACT_KEY_CAP")) {
        const long v = std::atol(e);
        if (v > 0) return (size_t)v;
    }
    return (size_t)1 << 18;
}
// ---- the search of a row-sharded problem (SURVEY 8e "shard by spatial cell") -----------------------------------------------
// The box list is built and sorted by every rank (a dozen small launches over the collision surface), but the SWEEP — the narrow phase of every
// candidate pair, the bulk of a search's kernel time — is dealt out: rank r runs every world-th workgroup of the sweep's grid and finds its share
// of the pairs. One all-gather brings every rank's keys (and its intersection count) to every rank; merged in rank order and sorted like the
// single rank's list, they give every rank the same tables. Per-rank capacity of the exchange: cs.shard_k keys (grown, and the search repeated,
// when a rank finds more).
bool search_is_sharded(const Context& c, const ContactSystem& cs) { return c.world > 1 && c.coll != nullptr && !c.no_sharded_search && !cs.brute_force; }
__global__ __launch_bounds__(CB) void k_shard_pack(const uint64_t* __restrict__ keys, const int* __restrict__ counters, int K, int key_cap, double* __restrict__ send)
{
    const int n = min(min(counters[0], key_cap), K);
    for (int i = blockIdx.x * CB + threadIdx.x; i < K + 2; i += gridDim.x * CB) {
        if (i == 0) send[0] = (double)counters[0];
        else if (i == 1) send[1] = (double)counters[1];
        else send[i] = i - 2 < n ? __longlong_as_double((long long)keys[i - 2]) : 0.0;
    }
}
// keys of all ranks, rank after rank, into the key list; counters[0] = their number, counters[1] = the ranks' intersection hits,
// counters[57] = 1 when a rank had more keys than fit its slot
__global__ __launch_bounds__(CB) void k_shard_merge(const double* __restrict__ recv, int W, int K, uint64_t* __restrict__ keys, int key_cap, int* __restrict__ counters)
{
    __shared__ int off[MAX_IPC_RANKS + 1];
    __shared__ int over;
    if (threadIdx.x == 0) {
        int at = 0, hits = 0, ov = 0;
        for (int r = 0; r < W; r++) {
            off[r] = at;
            const int cnt = (int)recv[(size_t)r * (K + 2)];
            hits += (int)recv[(size_t)r * (K + 2) + 1];
            if (cnt > K) ov = 1;
            at += cnt;
        }
        off[W] = at;
        over = ov;
        if (blockIdx.x == 0) {
            counters[0] = at;
            counters[1] = hits;
            counters[57] = ov;
        }
    }
    __syncthreads();
    if (over) return;
    for (int t = blockIdx.x * CB + threadIdx.x; t < W * K; t += gridDim.x * CB) {
        const int r = t / K, i = t - r * K;
        if (i < off[r + 1] - off[r] && off[r] + i < key_cap) keys[off[r] + i] = (uint64_t)__double_as_longlong(recv[(size_t)r * (K + 2) + 2 + i]);
    }
}
void merge_sharded_search(Context& c, ContactSystem& cs)
{
    if (!search_is_sharded(c, cs)) return;
    if (c.world > MAX_IPC_RANKS) throw Error("sharded contact search: too many ranks");
    if (cs.shard_k == 0) cs.shard_k = 8192;
    const int K = cs.shard_k, W = c.world;
    cs.shard_send.ensure((size_t)K + 2);
    cs.shard_recv.ensure((size_t)W * (K + 2));
    hipLaunchKernelGGL(k_shard_pack, dim3(std::min((K + 2 + CB - 1) / CB, 256)), dim3(CB), 0, c.stream, (const uint64_t*)cs.keys.p, (const int*)cs.counters.p, K, (int)cs.key_cap, cs.shard_send.p);
    c.coll->allgather_f64(cs.shard_send.p, cs.shard_recv.p, (size_t)K + 2, c.stream);
    hipLaunchKernelGGL(k_shard_merge, dim3(std::min((W * K + CB - 1) / CB, 256)), dim3(CB), 0, c.stream, (const double*)cs.shard_recv.p, W, K, cs.keys.p, (int)cs.key_cap, cs.counters.p);
    cs.n_sharded_searches++;
}
// after the read-back of a search's counters: a rank's keys did not fit its slot of the exchange -> larger slots, search again
bool sharded_search_overflowed(Context& c, ContactSystem& cs, const int* h)
{
    if (!search_is_sharded(c, cs) || !h[57]) return false;
    cs.shard_k = std::max(2 * cs.shard_k, 2 * (h[0] / std::max(c.world, 1)) + 1024);
    return true;
}
template <bool PROX, bool FR>
void launch_sweep(Context& c, ContactSystem& cs, const ContactDev& d, double enl2, bool reset_tasks = false)
{
    cs.sweep_tasks.ensure(3 * (size_t)SWEEP_TASK_CAP + 4);
    // (the task counter lives among the search's counters, which every caller zeroes before the search: counters[56]; a second sweep behind
    // the same fill — the speculative proximity search — resets it itself)
    int* task_count = cs.counters.p + 56;
    if (reset_tasks) fill_async(c.stream, task_count, 0, sizeof(int));
    const int pt_on = (int)(cs.pt_enabled && cs.n_t > 0), ee_on = (int)(cs.ee_enabled && cs.n_e > 1);
    const bool sharded = search_is_sharded(c, cs);
    const int W = sharded ? c.world : 1, me = sharded ? c.rank : 0;
    const int wgs = (cs.bp_cap + CB / SWEEP_SUB - 1) / (CB / SWEEP_SUB);
    hipLaunchKernelGGL((k_sweep<PROX, FR>), dim3((wgs + W - 1) / W), dim3(CB), 0, c.stream, d, cs.bands, cs.s_idx, (const float*)cs.s_aabb.p, (const float*)cs.s_lo.p,
                       (const int*)cs.seg.p, pt_on, ee_on, enl2, cs.keys.p, cs.counters.p, (int)cs.key_cap, task_count, cs.sweep_tasks.p, me, W);
    hipLaunchKernelGGL((k_sweep_tasks<PROX, FR>), dim3(SWEEP_TASK_WAVES / (CB / 64)), dim3(CB), 0, c.stream, d, cs.bands, cs.s_idx, (const float*)cs.s_aabb.p, (const int*)cs.seg.p, pt_on,
                       ee_on, enl2, cs.keys.p, cs.counters.p, (int)cs.key_cap, (const int*)task_count, (const int*)cs.sweep_tasks.p);
}
// padded length of the key sort for the next search
int padded_key_count(const ContactSystem& cs)
{
    int n_pad = 4096;
    while (n_pad < 2 * std::max(cs.n_last, 0) + 1024) n_pad *= 2;
    return (int)std::min<size_t>((size_t)n_pad, cs.key_cap);
}
// keys[0, n_sort) sorted (n_dev: the count on the device, the rest is padding), table boundaries and "same as the installed list" flag into
// counters[8..] / counters[2]; returns the sorted list
// The contact keys of a search (a few thousand, padded to n_sort) sorted in ONE launch: every workgroup stages all keys in LDS tile by tile and
// every thread counts the keys that sort before its own (equal keys — the padding — by position): its rank is where it goes. The library's radix
// sort of 64-bit keys took five launches for this.
constexpr int RANK_SORT_MAX = 16384, RANK_TILE = 4096;
__global__ __launch_bounds__(CB) void k_rank_sort_keys(const uint64_t* __restrict__ keys, int n, uint64_t* __restrict__ out)
{
    __shared__ uint64_t tile[RANK_TILE];
    const int e = blockIdx.x * CB + threadIdx.x;
    const uint64_t mine = e < n ? keys[e] : 0ull;
    int r = 0;
    for (int t0 = 0; t0 < n; t0 += RANK_TILE) {
        const int len = min(RANK_TILE, n - t0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += CB) tile[t] = keys[t0 + t];
        __syncthreads();
        // (keys before position e count when <=, keys behind it when <: equal keys keep their order)
        for (int u = 0; u < len; u++) {
            const uint64_t k = tile[u];
            r += (k < mine || (k == mine && t0 + u < e)) ? 1 : 0;
        }
    }
    if (e < n) out[r] = mine;
}
const uint64_t* sort_and_bound(Context& c, ContactSystem& cs, int n_sort, const int* n_dev, bool compare)
{
    hipcub::DoubleBuffer<uint64_t> dk(cs.keys.p, cs.keys_alt.p);
    // (the one-launch rank sort alone, beside the library's box sort, was measured in round 5: contact callbacks 12 -> 25 ms per 20 iterations —
    // every thread reads all n keys out of LDS; it stays part of option seg_sort only)
    if (n_sort > 1 && n_sort <= RANK_SORT_MAX && c.seg_sort) {
        hipLaunchKernelGGL(k_rank_sort_keys, dim3((n_sort + CB - 1) / CB), dim3(CB), 0, c.stream, (const uint64_t*)cs.keys.p, n_sort, cs.keys_alt.p);
        dk.selector = 1;
    } else if (n_sort > 1) {
        size_t tmp = 0;
        MS_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, dk, n_sort, 0, 64, c.stream));
        cs.cub_tmp.ensure(tmp);
        MS_CHECK(hipcub::DeviceRadixSort::SortKeys(cs.cub_tmp.p, tmp, dk, n_sort, 0, 64, c.stream));
    }
    const uint64_t* sorted = dk.Current();
    hipLaunchKernelGGL(k_table_bounds, dim3((n_sort + 1 + CB - 1) / CB), dim3(CB), 0, c.stream, sorted, n_sort, n_dev, compare ? cs.prev.p : sorted, compare ? (int)cs.n_prev : -1,
                       cs.counters.p + 8, cs.counters.p + 2);
    return sorted;
}
// Runs detection and installs the tables [t0, t1). Returns the number of rows.
int64_t detect_and_route(Context& c, double dt, bool friction)
{
    ContactSystem& cs = CS(c);
    if (cs.meshes.empty()) return 0;
    static const bool trace = getenv("MISTARK_CONTACT_TRACE") != nullptr;
    if (trace)
        std::fprintf(stderr, "[contact] detect fr=%d cache_valid=%d layout_dirty=%d cache_v=%llu data_v=%llu bp_v=%llu dt_same=%d\n", (int)friction, (int)cs.cache_valid, (int)c.layout_dirty,
                     (unsigned long long)cs.cache_version, (unsigned long long)c.data_version, (unsigned long long)cs.bp_version, (int)(cs.cache_dt == dt));
    if (friction) cs.cache_valid = false;
    else if (cs.cache_valid && !c.no_contact_cache && !c.layout_dirty && cs.cache_version == c.data_version && cs.cache_dt == dt) return cs.cache_n;
    if (!friction) {
        cs.n_searches++;
        if (cs.searched_version == c.data_version && cs.searched_dt == dt && !c.no_contact_cache) cs.n_repeated_searches++;
    }
    double tp = now_s();
    auto lap = [&](int k) {
        const double t = now_s();
        g_prof.t[k] += t - tp;
        tp = t;
    };
    g_prof.n++;
    prepare(c);
    upload_meshes(c, cs);
    const double enl = 2.0 * max_thickness(c, cs);
    const float enl_f = nextafterf((float)enl, INFINITY) + 1.1920929e-07f;  // (float)enl + eps (AABBs.cpp:38), rounded up
    ContactDev d = dev_view(c, cs);
    lap(0);
    bool boxes_current = cs.bp_valid && !cs.brute_force && !c.no_contact_cache && cs.bp_version == c.data_version && cs.bp_dt == dt && cs.bp_enl == enl_f;
    if (!boxes_current) update_vertices(c, cs, d, dt, enl_f);
    if (cs.key_cap == 0) {
        cs.key_cap = initial_key_cap();
        cs.keys.ensure(cs.key_cap);
        cs.keys_alt.ensure(cs.key_cap);
    }
    int h[64];
    int n = 0;
    const int t0 = friction ? N_CONTACT_TABLES : 0, t1 = friction ? N_TABLES : N_CONTACT_TABLES;
    const uint64_t* sorted = cs.keys.p;
    const bool compare = !friction && cs.n_prev >= 0;
    if (compare) cs.prev.ensure(std::max<size_t>((size_t)cs.n_prev, 1));
    const bool speculated = !friction && cs.spec.valid && !c.no_contact_cache && !cs.brute_force && cs.spec.version == c.data_version && cs.spec.dt == dt && cs.spec.enl == enl_f;
    if (speculated) {  // the intersection check of this very state has searched already
        std::memcpy(h, cs.spec.h, sizeof(h));
        n = cs.spec.n;
        sorted = cs.spec.sorted;
    }
    cs.spec.valid = false;  // (any search below reuses the key buffers)
    for (; !speculated;) {
        fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
        // One read-back per search: the key list is sorted over a padded length chosen from the previous search's count (padding keys sort
        // last and carry a table id no table has), the table boundaries are found with the count still on the device, and count,
        // boundaries and "same as before" flag come back together. A count beyond the padded length takes the two-step path below.
        const int n_pad = c.no_contact_cache ? 0 : padded_key_count(cs);
        if (!cs.brute_force) {
            if (!boxes_current) sort_boxes(c, cs, d);
            boxes_current = false;
            cs.bp_valid = true;
            cs.bp_version = c.data_version;
            cs.bp_dt = dt;
            cs.bp_enl = enl_f;
            if (friction) launch_sweep<true, true>(c, cs, d, enl * enl);
            else launch_sweep<true, false>(c, cs, d, enl * enl);
            merge_sharded_search(c, cs);
        } else {
        if (cs.pt_enabled && cs.n_t > 0) {
            const int chunk = chunk_for(cs.n_v, cs.n_t);
            const dim3 g((cs.n_v + CB - 1) / CB, (cs.n_t + chunk - 1) / chunk);
            if (friction) hipLaunchKernelGGL(k_detect_pt<true>, g, dim3(CB), 0, c.stream, d, enl * enl, chunk, cs.keys.p, cs.counters.p, (int)cs.key_cap);
            else hipLaunchKernelGGL(k_detect_pt<false>, g, dim3(CB), 0, c.stream, d, enl * enl, chunk, cs.keys.p, cs.counters.p, (int)cs.key_cap);
        }
        if (cs.ee_enabled && cs.n_e > 1) {
            const int chunk = chunk_for(cs.n_e, cs.n_e);
            const dim3 g((cs.n_e + CB - 1) / CB, (cs.n_e + chunk - 1) / chunk);
            if (friction) hipLaunchKernelGGL(k_detect_ee<true>, g, dim3(CB), 0, c.stream, d, enl * enl, chunk, cs.keys.p, cs.counters.p, (int)cs.key_cap);
            else hipLaunchKernelGGL(k_detect_ee<false>, g, dim3(CB), 0, c.stream, d, enl * enl, chunk, cs.keys.p, cs.counters.p, (int)cs.key_cap);
        }
        }
        if (n_pad > 0) {
            hipLaunchKernelGGL(k_pad_keys, dim3((n_pad + CB - 1) / CB), dim3(CB), 0, c.stream, cs.keys.p, (const int*)cs.counters.p, n_pad);
            sorted = sort_and_bound(c, cs, n_pad, (const int*)cs.counters.p, compare);
        }
        lap(1);
        fetch(c, h, cs.counters.p, 64 * sizeof(int));
        lap(2);
        n = h[0];
        if (sharded_search_overflowed(c, cs, h)) continue;
        if (!cs.brute_force && h[58] && cs.seg_sort_ok) {  // a segment of the box list too long for the in-LDS sort: the library sort from now on
            cs.seg_sort_ok = false;
            cs.bp_valid = false;
            continue;
        }
        if (!cs.brute_force && h[51] > cs.bp_cap) {  // (counters[48 + 3]) the banded box list did not fit: grow and search again
            cs.bp_cap = h[51] + h[51] / 4;
            cs.bp_valid = false;
            continue;
        }
        if ((size_t)n > cs.key_cap) {
            cs.key_cap = (size_t)n + n / 2;  // the list did not fit: grow and search again
            cs.keys.ensure(cs.key_cap);
            cs.keys_alt.ensure(cs.key_cap);
            cs.n_prev = -1;
            continue;
        }
        if (n_pad == 0 || n > n_pad) {  // count first, then sort exactly that many (first searches, bursts, option no_contact_cache)
            if (n_pad > 0) {  // the padded sort moved the first n_pad keys into the other buffer: search again with a longer padding
                cs.n_last = n;
                continue;
            }
            fill_async(c.stream, cs.counters.p + 2, 0, sizeof(int));
            sorted = sort_and_bound(c, cs, n, nullptr, compare);
            lap(3);
            fetch(c, h, cs.counters.p, 64 * sizeof(int));
            lap(4);
        }
        break;
    }
    cs.n_last = n;
    const int* bounds = h + 8;
    const bool unchanged = compare && n == cs.n_prev && h[2] == 0;
    if (trace) std::fprintf(stderr, "[contact]   searched: n=%d n_prev=%d differs=%d unchanged=%d speculated=%d\n", n, (int)cs.n_prev, h[2], (int)unchanged, (int)speculated);
    if (unchanged) {
        // the installed tables ARE this state's: the evaluation that opens the next Newton iteration asks again at the same state (round 5: the
        // cache used to be refreshed only by an installation, and every search that found the tables unchanged was followed by a second one)
        if (!friction) {
            cs.cache_valid = true;
            cs.cache_version = c.data_version;
            cs.cache_dt = dt;
            cs.cache_n = n;
            cs.searched_version = c.data_version;
            cs.searched_dt = dt;
        }
        return n;
    }
    // install the new row counts; buffers are (re)allocated before the routing kernel writes them
    if (!cs.td_pinned) MS_CHECK(hipHostMalloc((void**)&cs.td_pinned, N_TABLES * sizeof(TableDev)));
    TableDev* td = cs.td_pinned;  // pinned: the upload below is truly asynchronous and needs no trailing synchronisation
    std::memset(td, 0, N_TABLES * sizeof(TableDev));
    for (int t = t0; t < t1; t++) {
        ContactSystem::Table& T = cs.tables[t];
        const int rows = bounds[t + 1] - bounds[t];
        if (T.pot < 0) {
            if (rows) throw Error(std::string("contact: rows for table '") + TABLE_NAMES[t] + "' but its physical system was not bound at mistark_contact_init");
            continue;
        }
        Potential& P = c.pots[T.pot];
        if (rows != P.n_elem) c.layout_dirty = true;
        if (rows || P.n_elem) c.part[P.part].dirty = true;
        T.n = rows;
        P.n_elem = rows;
        T.conn.ensure(std::max<size_t>((size_t)rows * T.stride, 1));
        P.conn_ext = T.conn.p;
        for (int id : {T.a_T, T.a_mu, T.a_fn, T.a_bary})
            if (id >= 0 && c.arrays[id].n_items != rows) {
                c.arrays[id].n_items = rows;
                c.layout_dirty = true;
            }
    }
    c.layout_dirty = true;  // (conn_ext may have moved: refresh the kernels' argument blocks)
    prepare(c);
    for (int t = t0; t < t1; t++) {
        ContactSystem::Table& T = cs.tables[t];
        TableDev& o = td[t];
        o.conn = T.conn.p;
        o.stride = T.stride;
        o.nbary = T.nbary;
        o.start = bounds[t];
        o.T = T.a_T >= 0 ? c.arrays[T.a_T].dev : nullptr;
        o.mu = T.a_mu >= 0 ? c.arrays[T.a_mu].dev : nullptr;
        o.fn = T.a_fn >= 0 ? c.arrays[T.a_fn].dev : nullptr;
        o.bary = T.a_bary >= 0 ? c.arrays[T.a_bary].dev : nullptr;
    }
    lap(5);
    if (n > 0) {
        MS_CHECK(hipMemcpyAsync(cs.tables_dev.p, td, N_TABLES * sizeof(TableDev), hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(k_route, dim3((n + CB - 1) / CB), dim3(CB), 0, c.stream, d, sorted, n, (const TableDev*)cs.tables_dev.p, arr_dev(c, cs.arr.k));
    }
    if (!friction) {
        cs.prev.ensure(std::max<size_t>((size_t)n, 1));
        if (n > 0) copy_async(c.stream, cs.prev.p, sorted, (size_t)n * sizeof(uint64_t));
        cs.n_prev = n;
        cs.cache_valid = true;
        cs.cache_version = c.data_version;  // (after this function's own layout changes)
        cs.cache_dt = dt;
        cs.cache_n = n;
        cs.searched_version = c.data_version;
        cs.searched_dt = dt;
    }
    lap(6);
    return n;
}
int64_t count_intersections_uncached(Context& c, double dt);
// (the check of an accepted line-search candidate and the check of the converged state one evaluation later see the same DoFs: answered from
// the last count while nothing the vertices depend on has changed — Context::data_version)
int64_t count_intersections(Context& c, double dt)
{
    ContactSystem& cs = CS(c);
    static const bool trace = getenv("MISTARK_CONTACT_TRACE") != nullptr;
    if (trace) std::fprintf(stderr, "[contact] intersections: ix_valid=%d ix_v=%llu data_v=%llu bp_v=%llu\n", (int)cs.ix_valid, (unsigned long long)cs.ix_version, (unsigned long long)c.data_version, (unsigned long long)cs.bp_version);
    if (cs.ix_valid && !c.no_contact_cache && cs.ix_version == c.data_version && cs.ix_dt == dt) return cs.ix_n;
    const int64_t n = count_intersections_uncached(c, dt);
    cs.ix_valid = true;
    cs.ix_version = c.data_version;
    cs.ix_dt = dt;
    cs.ix_n = n;
    return n;
}
int64_t count_intersections_uncached(Context& c, double dt)
{
    ContactSystem& cs = CS(c);
    if (cs.meshes.empty() || cs.n_e == 0 || cs.n_t == 0) return 0;
    prepare(c);
    upload_meshes(c, cs);
    ContactDev d = dev_view(c, cs);
    // The boxes carry the enlargement of the proximity search (a superset of the candidates of tight boxes; the edge-triangle test itself
    // is exact): the proximity search of the evaluation that follows an accepted candidate then reuses the sorted list as it is.
    const double enl = 2.0 * max_thickness(c, cs);
    const float enl_f = cs.brute_force ? 0.f : nextafterf((float)enl, INFINITY) + 1.1920929e-07f;
    const bool boxes_current = cs.bp_valid && !cs.brute_force && !c.no_contact_cache && cs.bp_version == c.data_version && cs.bp_dt == dt && cs.bp_enl == enl_f;
    if (!boxes_current) update_vertices(c, cs, d, dt, enl_f);
    fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
    if (!cs.brute_force) {
        if (cs.key_cap == 0) {
            cs.key_cap = initial_key_cap();
            cs.keys.ensure(cs.key_cap);
            cs.keys_alt.ensure(cs.key_cap);
        }
        cs.spec.valid = false;
        const bool speculate = !c.no_contact_cache && c.contact_speculation;  // (option; off by default)
        for (bool first = true;; first = false) {
            if (!(first && boxes_current)) sort_boxes(c, cs, d);
            cs.bands.shrink = enl_f > 0.f ? 2.f * enl_f * (1.f - 1e-3f) : 0.f;
            launch_sweep<false, false>(c, cs, d, 0.0);
            // ... and, behind it, the proximity search of the evaluation that follows an accepted candidate (same state, same boxes), up
            // to the table boundaries: its counts come back with the intersection count, detect_and_route then only routes
            const uint64_t* sorted = nullptr;
            int n_pad = 0;
            if (speculate) {
                const bool compare = cs.n_prev >= 0;
                if (compare) cs.prev.ensure(std::max<size_t>((size_t)cs.n_prev, 1));
                launch_sweep<true, false>(c, cs, d, enl * enl, /*reset_tasks=*/true);
                merge_sharded_search(c, cs);
                n_pad = padded_key_count(cs);
                hipLaunchKernelGGL(k_pad_keys, dim3((n_pad + CB - 1) / CB), dim3(CB), 0, c.stream, cs.keys.p, (const int*)cs.counters.p, n_pad);
                sorted = sort_and_bound(c, cs, n_pad, (const int*)cs.counters.p, compare);
            }
            if (!speculate) merge_sharded_search(c, cs);  // (the ranks' intersection counts; with speculation the merge above carried them)
            int hb[64];
            fetch(c, hb, cs.counters.p, sizeof(hb));
            if (sharded_search_overflowed(c, cs, hb)) {
                fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
                continue;
            }
            if (hb[58] && cs.seg_sort_ok) {
                cs.seg_sort_ok = false;
                cs.bp_valid = false;
                fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
                continue;
            }
            if (hb[51] > cs.bp_cap) {
                cs.bp_cap = hb[51] + hb[51] / 4;
                cs.bp_valid = false;
                fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
                continue;
            }
            cs.bp_valid = true;
            cs.bp_version = c.data_version;
            cs.bp_dt = dt;
            cs.bp_enl = enl_f;
            if (speculate && (size_t)hb[0] <= cs.key_cap && hb[0] <= n_pad) {
                cs.spec.valid = true;
                cs.spec.version = c.data_version;
                cs.spec.dt = dt;
                cs.spec.enl = enl_f;
                cs.spec.n = hb[0];
                cs.spec.sorted = sorted;
                std::memcpy(cs.spec.h, hb, sizeof(hb));
            }
            if (speculate) cs.n_last = hb[0];
            return hb[1];
        }
    } else {
        const int chunk = chunk_for(cs.n_e, cs.n_t);
        hipLaunchKernelGGL(k_detect_et, dim3((cs.n_e + CB - 1) / CB, (cs.n_t + chunk - 1) / chunk), dim3(CB), 0, c.stream, d, chunk, cs.counters.p);
    }
    int h[2];
    fetch(c, h, cs.counters.p, sizeof(h));
    return h[1];
}
int find_table(const char* name)
{
    for (int t = 0; t < N_TABLES; t++)
        if (std::strcmp(name, TABLE_NAMES[t]) == 0) return t;
    return -1;
}
}  // namespace

// ---- the detector as a standalone service (include/mistark_tmcd.h): host positions in, the reference's result lists out -----------------------
// One private context with a contact system that has no tables: every mesh counts as deformable and nothing is filtered by thickness, so the
// six barrier tables of the deformable family ARE tmcd's six lists (table_of(family, 0, 0)); the sorted keys come back to the host and are
// expanded into rows there.
// A host buffer the device transfers into / out of directly: page-locked (hipHostMalloc), grown geometrically, contents NOT kept across a
// growth. The standalone detector gathers the caller's positions into one (the upload is then a single DMA instead of a copy through HIP's
// staging buffer) and reads its key list back into another.
template <class T>
struct PinnedBuf
{
    T* p = nullptr;
    size_t n = 0, cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    void resize(size_t m)
    {
        if (m > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = std::max(m + m / 4, (size_t)1024);
            MS_CHECK(hipHostMalloc((void**)&p, cap * sizeof(T)));
        }
        n = m;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* begin() { return p; }
    T* end() { return p + n; }
};
struct StandaloneDetector
{
    Context c;
    std::vector<const double*> xm;
    PinnedBuf<double> X;
    std::vector<int32_t> rows[6], et_rows, bp_rows[2];  // (bp_rows: the last broad-phase listing, point-triangle | edge-edge, 4 columns)
    std::vector<double> dist[6];
    PinnedBuf<uint64_t> keys;
    std::string last_error;
    // A caller asks again at unchanged positions more often than not (the reference's contact class searches at every energy evaluation: the
    // accepted line-search candidate's evaluation and the evaluation that opens the next Newton iteration see the same vertices): the
    // lists of the previous run are the answer when positions, enlargement, activation and the mesh set are the same bytes.
    uint64_t prox_print = 0, et_print = 0;
    bool prox_valid = false, et_valid = false;
    int32_t et_count = 0;
    int64_t n_prox_cached = 0, n_et_cached = 0;
};
namespace {
constexpr int CD_COLS[6] = {8, 9, 7, 10, 9, 8};
uint64_t cd_fingerprint(const StandaloneDetector& D, const ContactSystem& cs, double enlargement)
{
    auto mix = [](uint64_t h, uint64_t v) {
        h ^= v;
        h *= 0x9E3779B97F4A7C15ull;
        return h ^ (h >> 29);
    };
    uint64_t h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    const uint64_t* w = reinterpret_cast<const uint64_t*>(D.X.data());
    const size_t n = D.X.size();
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int k = 0; k < 4; k++) h[k] = mix(h[k], w[i + k]);
    for (; i < n; i++) h[0] = mix(h[0], w[i]);
    uint64_t e;
    std::memcpy(&e, &enlargement, 8);
    uint64_t r = mix(mix(mix(mix(h[0], h[1]), h[2]), h[3]), e);
    r = mix(r, (uint64_t)cs.meshes.size() * 4 + (cs.pt_enabled ? 2 : 0) + (cs.ee_enabled ? 1 : 0));
    r = mix(r, (uint64_t)cs.disabled_pairs.size() + 1000003ull * (uint64_t)(cs.h_bl_pt.size() + cs.h_bl_ee.size()));
    return r;
}
void cd_gather_positions(StandaloneDetector& D)
{
    ContactSystem& cs = CS(D.c);
    D.X.resize(3 * (size_t)cs.n_v);
    for (size_t g = 0; g < cs.meshes.size(); g++) std::memcpy(D.X.data() + 3 * (size_t)cs.meshes[g].v_off, D.xm[g], 3 * (size_t)cs.meshes[g].n_v * sizeof(double));
}
void cd_upload_positions(StandaloneDetector& D)
{
    ContactSystem& cs = CS(D.c);
    cs.X.ensure(std::max<size_t>(D.X.size(), 1));
    cs.aabb.ensure(6 * (size_t)std::max(cs.n_v + cs.n_t + cs.n_e, 1));
    if (!D.X.empty()) MS_CHECK(hipMemcpyAsync(cs.X.p, D.X.data(), D.X.size() * sizeof(double), hipMemcpyHostToDevice, D.c.stream));
}
ContactDev cd_view(StandaloneDetector& D)
{
    Context& c = D.c;
    ContactSystem& cs = CS(c);
    upload_meshes(c, cs);
    if (cs.thick_override.cap < cs.meshes.size()) {
        const std::vector<double> big(cs.meshes.size(), 1e300);
        upload(c, cs.thick_override, big);
        MS_CHECK(hipStreamSynchronize(c.stream));
    }
    return dev_view(c, cs);
}
// searches until the box list and the key list fit; returns the number of keys (sorted when `sort`), h = the counters
int cd_search(StandaloneDetector& D, const ContactDev& d, bool proximity, double enl, int* h)
{
    Context& c = D.c;
    ContactSystem& cs = CS(c);
    if (cs.key_cap == 0) {
        cs.key_cap = initial_key_cap();
        cs.keys.ensure(cs.key_cap);
        cs.keys_alt.ensure(cs.key_cap);
    }
    for (;;) {
        fill_async(c.stream, cs.counters.p, 0, 64 * sizeof(int));
        sort_boxes(c, cs, d);
        if (proximity) launch_sweep<true, false>(c, cs, d, enl * enl);
        else {
            cs.bands.shrink = 0.f;  // (tight boxes: mistark_cd_run_intersection builds its own)
            launch_sweep<false, false>(c, cs, d, -1.0);
        }
        fetch(c, h, cs.counters.p, 64 * sizeof(int));
        if (h[58] && cs.seg_sort_ok) {
            cs.seg_sort_ok = false;
            continue;
        }
        if (h[51] > cs.bp_cap) {
            cs.bp_cap = h[51] + h[51] / 4;
            continue;
        }
        if ((size_t)h[0] > cs.key_cap) {
            cs.key_cap = (size_t)h[0] + h[0] / 2;
            cs.keys.ensure(cs.key_cap);
            cs.keys_alt.ensure(cs.key_cap);
            continue;
        }
        return h[0];
    }
}
}  // namespace
}  // namespace mistark

using namespace mistark;
struct mistark_cd
{
    StandaloneDetector D;
};
#define CD_BEGIN          \
    if (!cd) return -1;   \
    try {
#define CD_END(ret)                      \
    }                                    \
    catch (const std::exception& e)      \
    {                                    \
        cd->D.last_error = e.what();     \
        return -1;                       \
    }                                    \
    return ret;
extern "C" {
int mistark_cd_create(mistark_cd** out, int device)
{
    if (!out) return -1;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return -3;  // (no GPU: there is no host detector to fall back to)
    if (device < 0 || device >= n_dev) return -2;
    if (hipSetDevice(device) != hipSuccess) return -4;
    mistark_cd* cd = new mistark_cd();
    Context& c = cd->D.c;
    c.device = device;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
        delete cd;
        return -5;
    }
    auto* cs = new ContactSystem();
    c.contact = cs;
    const int32_t none[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    std::memcpy(&cs->arr, none, sizeof(none));
    cs->counters.ensure(64);
    *out = cd;
    return 0;
}
void mistark_cd_destroy(mistark_cd* cd)
{
    if (!cd) return;
    (void)hipSetDevice(cd->D.c.device);
    (void)hipStreamSynchronize(cd->D.c.stream);
    delete cd;  // (Context's destructor releases the contact system, the buffers and the stream)
}
const char* mistark_cd_last_error(mistark_cd* cd) { return cd ? cd->D.last_error.c_str() : "null detector"; }
int mistark_cd_add_mesh(mistark_cd* cd, const double* xm, int32_t n_vertices, const int32_t* triangles, int32_t n_triangles, const int32_t* edges, int32_t n_edges)
{
    int g = -1;
    CD_BEGIN
    if (!xm || (n_triangles > 0 && !triangles) || (n_edges > 0 && !edges)) throw Error("cd: null mesh data");
    std::vector<int32_t> src((size_t)std::max(n_vertices, 0));
    for (int i = 0; i < n_vertices; i++) src[i] = i;
    g = contact_add_mesh(cd->D.c, MISTARK_CONTACT_DEFORMABLE, 0, src.data(), n_vertices, triangles, n_triangles, edges, n_edges);
    cd->D.xm.push_back(xm);
    CD_END(g)
}
int mistark_cd_add_blacklist(mistark_cd* cd, int32_t a, int32_t b)
{
    CD_BEGIN
    ContactSystem& cs = CS(cd->D.c);
    if (a < 0 || b < 0 || a >= (int)cs.meshes.size() || b >= (int)cs.meshes.size()) throw Error("cd: bad mesh id");
    cs.disabled_pairs.push_back({a, b});
    cs.meshes_dirty = true;
    CD_END(0)
}
int mistark_cd_add_blacklist_range(mistark_cd* cd, int32_t edge_edge, int32_t mesh_a, int32_t a0, int32_t a1, int32_t mesh_b, int32_t b0, int32_t b1)
{
    // tmcd::ProximityDetection::add_blacklist_range_point_triangle (edge_edge = 0: points [a0, a1) of mesh_a against triangles [b0, b1) of mesh_b) and
    // add_blacklist_range_edge_edge (edge_edge = 1: edges [a0, a1) of mesh_a against edges [b0, b1) of mesh_b; the first interval must be the lower
    // one in the global edge numbering, as in the reference, BroadPhasePTEEBase.cpp:37-40)
    CD_BEGIN
    ContactSystem& cs = CS(cd->D.c);
    const int nm = (int)cs.meshes.size();
    if (mesh_a < 0 || mesh_b < 0 || mesh_a >= nm || mesh_b >= nm) throw Error("cd: bad mesh id");
    const ContactSystem::Mesh &A = cs.meshes[(size_t)mesh_a], &B = cs.meshes[(size_t)mesh_b];
    if (edge_edge) {
        if (a0 < 0 || a1 > A.n_e || a0 > a1 || b0 < 0 || b1 > B.n_e || b0 > b1) throw Error("cd: edge interval outside its mesh");
        if (A.e_off + a0 > B.e_off + b0) throw Error("cd: add_blacklist_range_edge_edge: the first edge interval must be the lower one (as in the reference)");
        for (int v : {A.e_off + a0, A.e_off + a1, B.e_off + b0, B.e_off + b1}) cs.h_bl_ee.push_back(v);
    } else {
        if (a0 < 0 || a1 > A.n_v || a0 > a1 || b0 < 0 || b1 > B.n_t || b0 > b1) throw Error("cd: point / triangle interval outside its mesh");
        for (int v : {B.t_off + b0, B.t_off + b1, A.v_off + a0, A.v_off + a1}) cs.h_bl_pt.push_back(v);
    }
    cs.meshes_dirty = true;
    cd->D.prox_valid = false;
    CD_END(0)
}
int mistark_cd_activate(mistark_cd* cd, int point_triangle, int edge_edge)
{
    CD_BEGIN
    ContactSystem& cs = CS(cd->D.c);
    cs.pt_enabled = point_triangle != 0;
    cs.ee_enabled = edge_edge != 0;
    CD_END(0)
}
int mistark_cd_run_proximity(mistark_cd* cd, double enlargement, int32_t counts[6])
{
    CD_BEGIN
    StandaloneDetector& D = cd->D;
    Context& c = D.c;
    ContactSystem& cs = CS(c);
    MS_CHECK(hipSetDevice(c.device));
    if (cs.meshes.empty()) {
        for (int l = 0; l < 6; l++) {
            D.rows[l].clear();
            D.dist[l].clear();
            if (counts) counts[l] = 0;
        }
        return 0;
    }
    if (!(enlargement >= 0.0)) throw Error("cd: negative enlargement");
    if (cs.meshes_dirty) upload_meshes(c, cs);  // (n_v of a mesh added since the last run)
    cd_gather_positions(D);
    const uint64_t print = cd_fingerprint(D, cs, enlargement);
    static const bool no_cache = std::getenv("MISTARK_CD_NO_CACHE") != nullptr;
    if (D.prox_valid && print == D.prox_print && !no_cache) {  // the same question as last time: the same lists
        for (int l = 0; l < 6; l++)
            if (counts) counts[l] = (int32_t)(D.rows[l].size() / (size_t)CD_COLS[l]);
        D.n_prox_cached++;
        return 0;
    }
    D.prox_valid = false;
    for (int l = 0; l < 6; l++) {
        D.rows[l].clear();
        D.dist[l].clear();
        if (counts) counts[l] = 0;
    }
    cd_upload_positions(D);
    const ContactDev d = cd_view(D);
    D.prox_print = print;
    const float enl_f = nextafterf((float)enlargement, INFINITY) + 1.1920929e-07f;  // (float)enl + eps (AABBs.cpp:38), rounded up
    const int np = cs.n_v + cs.n_t + cs.n_e;
    hipLaunchKernelGGL(k_contact_aabbs, dim3((np + CB - 1) / CB), dim3(CB), 0, c.stream, d, enl_f, cs.aabb.p);
    int h[64];
    const int n = cd_search(D, d, true, enlargement, h);
    if (n == 0) {
        D.prox_valid = true;
        return 0;
    }
    fill_async(c.stream, cs.counters.p + 2, 0, sizeof(int));
    const uint64_t* sorted = sort_and_bound(c, cs, n, nullptr, false);
    D.keys.resize((size_t)n);
    MS_CHECK(hipMemcpyAsync(D.keys.data(), sorted, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
    fetch(c, h, cs.counters.p, 64 * sizeof(int));
    const int* bounds = h + 8;
    const uint64_t pmask = (1ull << PRIM_BITS) - 1;
    auto X3 = [&](int v) { return d3(D.X[3 * (size_t)v], D.X[3 * (size_t)v + 1], D.X[3 * (size_t)v + 2]); };
    for (int l = 0; l < 6; l++) {
        const int k0 = bounds[l], k1 = bounds[l + 1], cols = CD_COLS[l];
        D.rows[l].resize((size_t)(k1 - k0) * cols);
        D.dist[l].resize((size_t)(k1 - k0));
        if (counts) counts[l] = k1 - k0;
        for (int k = k0; k < k1; k++) {
            const uint64_t key = D.keys[(size_t)k];
            const int type = (int)((key >> 53) & 0xF), a = (int)((key >> PRIM_BITS) & pmask), b = (int)(key & pmask);
            int32_t* r = D.rows[l].data() + (size_t)(k - k0) * cols;
            int ty;
            if (l < 3) {  // (point a, triangle b): ProximityDetection.cpp:113-129
                const int mp = cs.h_cv_mesh[a], mt = cs.h_tri_mesh[b];
                const ContactSystem::Mesh &Mp = cs.meshes[mp], &Mt = cs.meshes[mt];
                const int t[3] = {cs.h_tri[3 * (size_t)b], cs.h_tri[3 * (size_t)b + 1], cs.h_tri[3 * (size_t)b + 2]};
                r[0] = mp; r[1] = a - Mp.v_off;
                r[2] = mt; r[3] = b - Mt.t_off; r[4] = t[0] - Mt.v_off; r[5] = t[1] - Mt.v_off; r[6] = t[2] - Mt.v_off;
                if (l == 0) r[7] = t[type - P_T0] - Mt.v_off;
                if (l == 1) {
                    r[7] = t[type - P_E0] - Mt.v_off;
                    r[8] = t[(type - P_E0 + 1) % 3] - Mt.v_off;
                }
                D.dist[l][(size_t)(k - k0)] = std::sqrt(point_triangle_sq_distance(ty, X3(a), X3(t[0]), X3(t[1]), X3(t[2])));
            } else {  // (edge a, edge b): ProximityDetection.cpp:166-186
                const int ma = cs.h_edge_mesh[a], mb = cs.h_edge_mesh[b];
                const ContactSystem::Mesh &Ma = cs.meshes[ma], &Mb = cs.meshes[mb];
                const int ea[2] = {cs.h_edge[2 * (size_t)a], cs.h_edge[2 * (size_t)a + 1]}, eb[2] = {cs.h_edge[2 * (size_t)b], cs.h_edge[2 * (size_t)b + 1]};
                const int32_t EA[4] = {ma, a - Ma.e_off, ea[0] - Ma.v_off, ea[1] - Ma.v_off}, EB[4] = {mb, b - Mb.e_off, eb[0] - Mb.v_off, eb[1] - Mb.v_off};
                auto put = [&](int32_t* o, const int32_t* E, int point) {
                    for (int i = 0; i < 4; i++) o[i] = E[i];
                    if (point >= 0) o[4] = E[2 + point];
                };
                if (l == 3) {
                    put(r, EA, type == EA0_EB0 || type == EA0_EB1 ? 0 : 1);
                    put(r + 5, EB, type == EA0_EB0 || type == EA1_EB0 ? 0 : 1);
                } else if (l == 4) {
                    if (type == EA_EB0 || type == EA_EB1) {  // the point is on edge b: it comes first
                        put(r, EB, type == EA_EB0 ? 0 : 1);
                        put(r + 5, EA, -1);
                    } else {
                        put(r, EA, type == EA0_EB ? 0 : 1);
                        put(r + 5, EB, -1);
                    }
                } else {
                    put(r, EA, -1);
                    put(r + 4, EB, -1);
                }
                D.dist[l][(size_t)(k - k0)] = std::sqrt(edge_edge_sq_distance(ty, X3(ea[0]), X3(ea[1]), X3(eb[0]), X3(eb[1])));
            }
        }
    }
    D.prox_valid = true;
    CD_END(0)
}
int mistark_cd_get_proximity(mistark_cd* cd, int list, int32_t* rows, double* distance)
{
    CD_BEGIN
    if (list < 0 || list >= 6) throw Error("cd: bad list");
    if (rows && !cd->D.rows[list].empty()) std::memcpy(rows, cd->D.rows[list].data(), cd->D.rows[list].size() * sizeof(int32_t));
    if (distance && !cd->D.dist[list].empty()) std::memcpy(distance, cd->D.dist[list].data(), cd->D.dist[list].size() * sizeof(double));
    CD_END(0)
}
int mistark_cd_run_broad_phase(mistark_cd* cd, double enlargement, int32_t counts[2])
{
    CD_BEGIN
    StandaloneDetector& D = cd->D;
    Context& c = D.c;
    ContactSystem& cs = CS(c);
    MS_CHECK(hipSetDevice(c.device));
    for (int l = 0; l < 2; l++) {
        D.bp_rows[l].clear();
        if (counts) counts[l] = 0;
    }
    if (cs.meshes.empty()) return 0;
    if (!(enlargement >= 0.0)) throw Error("cd: negative enlargement");
    if (cs.meshes_dirty) upload_meshes(c, cs);
    cd_gather_positions(D);
    cd_upload_positions(D);
    ContactDev d = cd_view(D);
    d.broad_only = 1;
    d.broad_extra = (float)enlargement + 1.1920929e-07f;  // AABBs.cpp:38
    // the search itself runs on the engine's outward-rounded boxes (a superset: every pair the reference's boxes accept is among its candidates)
    const float enl_f = nextafterf((float)enlargement, INFINITY) + 1.1920929e-07f;
    const int np = cs.n_v + cs.n_t + cs.n_e;
    hipLaunchKernelGGL(k_contact_aabbs, dim3((np + CB - 1) / CB), dim3(CB), 0, c.stream, d, enl_f, cs.aabb.p);
    int h[64];
    const int n = cd_search(D, d, true, enlargement, h);
    D.prox_valid = false;  // (the key buffers and counters of the proximity lists were reused)
    if (n == 0) return 0;
    D.keys.resize((size_t)n);
    MS_CHECK(hipMemcpyAsync(D.keys.data(), cs.keys.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    std::sort(D.keys.begin(), D.keys.end());
    const uint64_t pmask = (1ull << PRIM_BITS) - 1;
    for (int k = 0; k < n; k++) {
        const uint64_t key = D.keys[(size_t)k];
        const int l = (int)(key >> 58), a = (int)((key >> PRIM_BITS) & pmask), b = (int)(key & pmask);
        if (l == 0) {  // {point.set, point.idx}, {triangle.set, triangle.idx}
            const int mp = cs.h_cv_mesh[a], mt = cs.h_tri_mesh[b];
            for (int v : {mp, a - cs.meshes[mp].v_off, mt, b - cs.meshes[mt].t_off}) D.bp_rows[0].push_back(v);
        } else {       // {edge_a.set, edge_a.idx}, {edge_b.set, edge_b.idx}, a before b in the global edge order
            const int ma = cs.h_edge_mesh[a], mb = cs.h_edge_mesh[b];
            for (int v : {ma, a - cs.meshes[ma].e_off, mb, b - cs.meshes[mb].e_off}) D.bp_rows[1].push_back(v);
        }
    }
    for (int l = 0; l < 2; l++)
        if (counts) counts[l] = (int32_t)(D.bp_rows[l].size() / 4);
    CD_END(0)
}
int mistark_cd_get_broad_phase(mistark_cd* cd, int list, int32_t* rows)
{
    CD_BEGIN
    if (list < 0 || list >= 2) throw Error("cd: bad broad-phase list");
    if (rows && !cd->D.bp_rows[list].empty()) std::memcpy(rows, cd->D.bp_rows[list].data(), cd->D.bp_rows[list].size() * sizeof(int32_t));
    CD_END(0)
}
int mistark_cd_run_intersection(mistark_cd* cd, int32_t* n_pairs)
{
    CD_BEGIN
    StandaloneDetector& D = cd->D;
    Context& c = D.c;
    ContactSystem& cs = CS(c);
    MS_CHECK(hipSetDevice(c.device));
    if (n_pairs) *n_pairs = 0;
    if (cs.meshes_dirty) upload_meshes(c, cs);
    if (cs.meshes.empty() || cs.n_e == 0 || cs.n_t == 0) {
        D.et_rows.clear();
        return 0;
    }
    cd_gather_positions(D);
    const uint64_t print = cd_fingerprint(D, cs, -1.0);
    static const bool no_cache = std::getenv("MISTARK_CD_NO_CACHE") != nullptr;
    if (D.et_valid && print == D.et_print && !no_cache) {
        if (n_pairs) *n_pairs = D.et_count;
        D.n_et_cached++;
        return 0;
    }
    D.et_valid = false;
    D.et_rows.clear();
    D.et_print = print;
    D.et_count = 0;
    cd_upload_positions(D);
    const ContactDev d = cd_view(D);
    const int np = cs.n_v + cs.n_t + cs.n_e;
    hipLaunchKernelGGL(k_contact_aabbs, dim3((np + CB - 1) / CB), dim3(CB), 0, c.stream, d, 0.f, cs.aabb.p);  // tight boxes (BroadPhaseET)
    int h[64];
    const int n = cd_search(D, d, false, 0.0, h);
    if (n != h[1]) throw Error("cd: intersection list and count disagree");
    if (n == 0) {
        D.et_valid = true;
        return 0;
    }
    D.keys.resize((size_t)n);
    MS_CHECK(hipMemcpyAsync(D.keys.data(), cs.keys.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
    MS_CHECK(hipStreamSynchronize(c.stream));
    std::sort(D.keys.begin(), D.keys.end());
    const uint64_t pmask = (1ull << PRIM_BITS) - 1;
    D.et_rows.resize(9 * (size_t)n);
    for (int k = 0; k < n; k++) {
        const int e = (int)((D.keys[(size_t)k] >> PRIM_BITS) & pmask), t = (int)(D.keys[(size_t)k] & pmask);
        const int me = cs.h_edge_mesh[e], mt = cs.h_tri_mesh[t];
        const ContactSystem::Mesh &Me = cs.meshes[me], &Mt = cs.meshes[mt];
        int32_t* r = D.et_rows.data() + 9 * (size_t)k;
        r[0] = me; r[1] = e - Me.e_off; r[2] = cs.h_edge[2 * (size_t)e] - Me.v_off; r[3] = cs.h_edge[2 * (size_t)e + 1] - Me.v_off;
        r[4] = mt; r[5] = t - Mt.t_off;
        for (int i = 0; i < 3; i++) r[6 + i] = cs.h_tri[3 * (size_t)t + i] - Mt.v_off;
    }
    if (n_pairs) *n_pairs = n;
    D.et_count = n;
    D.et_valid = true;
    CD_END(0)
}
int mistark_cd_get_intersections(mistark_cd* cd, int32_t* rows)
{
    CD_BEGIN
    if (rows && !cd->D.et_rows.empty()) std::memcpy(rows, cd->D.et_rows.data(), cd->D.et_rows.size() * sizeof(int32_t));
    CD_END(0)
}
}  // extern "C"


#define CAPI_BEGIN       \
    if (!ctx) return -1; \
    ::mistark::DryScope _dry(ctx->c.dry); \
    try {
#define CAPI_END(ret)                 \
    }                                 \
    catch (const std::exception& e)   \
    {                                 \
        ctx->c.last_error = e.what(); \
        return -1;                    \
    }                                 \
    return ret;

extern "C" {
int mistark_contact_init(mistark_ctx* ctx, const mistark_contact_arrays* arrays)
{
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    if (!arrays) throw Error("contact: null arrays");
    const int32_t* ids = &arrays->v1;
    for (int i = 0; i < 12; i++)
        if (ids[i] >= (int)ctx->c.arrays.size()) throw Error("contact: bad array id");
    contact_init(ctx->c, *arrays);
    CAPI_END(0)
}
int mistark_contact_add_mesh(mistark_ctx* ctx, int kind, int idx_in_ps, const int32_t* vertex_index, int32_t n_vertices, const int32_t* triangles, int32_t n_triangles,
                             const int32_t* edges, int32_t n_edges)
{
    int g = -1;
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    g = contact_add_mesh(ctx->c, kind, idx_in_ps, vertex_index, n_vertices, triangles, n_triangles, edges, n_edges);
    CAPI_END(g)
}
int mistark_contact_set_friction(mistark_ctx* ctx, int a, int b, double mu)
{
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    ContactSystem& cs = CS(ctx->c);
    const int nm = (int)cs.meshes.size();
    if (a < 0 || b < 0 || a >= nm || b >= nm) throw Error("contact: bad group id");
    cs.friction[{std::min(a, b), std::max(a, b)}] = mu;
    cs.meshes_dirty = true;
    CAPI_END(0)
}
int mistark_contact_disable_collision(mistark_ctx* ctx, int a, int b)
{
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    ContactSystem& cs = CS(ctx->c);
    const int nm = (int)cs.meshes.size();
    if (a < 0 || b < 0 || a >= nm || b >= nm) throw Error("contact: bad group id");
    cs.disabled_pairs.push_back({std::min(a, b), std::max(a, b)});
    cs.meshes_dirty = true;
    cs.n_prev = -1;
    CAPI_END(0)
}
int mistark_contact_set_broad_phase(mistark_ctx* ctx, int brute_force)
{
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    ContactSystem& cs = CS(ctx->c);
    cs.brute_force = brute_force != 0;
    cs.n_prev = -1;
    CAPI_END(0)
}
int mistark_contact_enable(mistark_ctx* ctx, int point_triangle, int edge_edge)
{
    CAPI_BEGIN
    ctx->c.touch();  // (invalidates the cached detection)
    ContactSystem& cs = CS(ctx->c);
    cs.pt_enabled = point_triangle != 0;
    cs.ee_enabled = edge_edge != 0;
    cs.n_prev = -1;
    CAPI_END(0)
}
int mistark_contact_update(mistark_ctx* ctx, double dt, int64_t* n_contacts)
{
    CAPI_BEGIN
    const int64_t n = detect_and_route(ctx->c, dt, false);
    if (n_contacts) *n_contacts = n;
    CAPI_END(0)
}
int mistark_contact_update_friction(mistark_ctx* ctx, int64_t* n_contacts)
{
    CAPI_BEGIN
    const int64_t n = detect_and_route(ctx->c, 0.0, true);
    if (n_contacts) *n_contacts = n;
    CAPI_END(0)
}
int mistark_contact_count_intersections(mistark_ctx* ctx, double dt, int64_t* n_found)
{
    CAPI_BEGIN
    const int64_t n = count_intersections(ctx->c, dt);
    if (n_found) *n_found = n;
    CAPI_END(0)
}
int mistark_contact_get_table(mistark_ctx* ctx, const char* potential, int32_t* conn, int32_t* n_rows, int32_t* stride)
{
    CAPI_BEGIN
    ContactSystem& cs = CS(ctx->c);
    const int t = find_table(potential);
    if (t < 0) throw Error(std::string("contact: unknown table '") + potential + "'");
    ContactSystem::Table& T = cs.tables[t];
    if (n_rows) *n_rows = T.n;
    if (stride) *stride = T.stride;
    if (conn && T.n > 0) {
        MS_CHECK(hipMemcpyAsync(conn, T.conn.p, (size_t)T.n * T.stride * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->c.stream));
        MS_CHECK(hipStreamSynchronize(ctx->c.stream));
    }
    CAPI_END(0)
}
int mistark_contact_get_friction_data(mistark_ctx* ctx, const char* potential, double* T_out, double* mu, double* fn, double* bary, int32_t* nbary)
{
    CAPI_BEGIN
    Context& c = ctx->c;
    ContactSystem& cs = CS(c);
    const int t = find_table(potential);
    if (t < N_CONTACT_TABLES) throw Error(std::string("contact: '") + potential + "' is not a friction table");
    ContactSystem::Table& T = cs.tables[t];
    if (nbary) *nbary = T.nbary;
    auto get = [&](int id, double* out, int stride) {
        if (out && id >= 0 && T.n > 0) MS_CHECK(hipMemcpyAsync(out, c.arrays[id].dev, (size_t)T.n * stride * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    };
    get(T.a_T, T_out, 6);
    get(T.a_mu, mu, 1);
    get(T.a_fn, fn, 1);
    get(T.a_bary, bary, T.nbary);
    MS_CHECK(hipStreamSynchronize(c.stream));
    CAPI_END(0)
}
int mistark_contact_get_vertices(mistark_ctx* ctx, double* x, int64_t* n_vertices)
{
    CAPI_BEGIN
    ContactSystem& cs = CS(ctx->c);
    if (n_vertices) *n_vertices = cs.n_v;
    if (x && cs.n_v > 0 && cs.X.p) {
        MS_CHECK(hipMemcpyAsync(x, cs.X.p, 3 * (size_t)cs.n_v * sizeof(double), hipMemcpyDeviceToHost, ctx->c.stream));
        MS_CHECK(hipStreamSynchronize(ctx->c.stream));
    }
    CAPI_END(0)
}
int mistark_contact_recipe(const char* potential, int32_t* conn_stride, int32_t* roles, int32_t* strides, int32_t* conn_cols)
{
    const int t = find_table(potential);
    if (t < 0) return -1;
    int stride = 0;
    const std::vector<Bind> r = recipe(t, stride);
    if (conn_stride) *conn_stride = stride;
    for (size_t i = 0; i < r.size(); i++) {
        if (roles) roles[i] = r[i].role;
        if (strides) strides[i] = r[i].stride;
        if (conn_cols) conn_cols[i] = r[i].col;
    }
    return (int)r.size();
}
}
