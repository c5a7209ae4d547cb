// host/attachments.cpp — rods and attachments of the host layer (see sim.hpp): stark::EnergySegmentStrain, stark::EnergyAttachments and
// the Line presets. The energies themselves are the device potentials of the same registry names (csrc/energies.hpp).
#include <limits>
#include <stdexcept>

#include "bind.hpp"
#include "sim.hpp"

namespace mistark {

// ======================================================================================================================
// EnergySegmentStrain  (stark/src/models/deformables/line/EnergySegmentStrain.cpp)
// ======================================================================================================================
EnergySegmentStrain::EnergySegmentStrain(Stark& s, spPointDynamics d) : stark(s), dyn(d) { stark.add_model(this); }
void EnergySegmentStrain::register_potentials(mistark_ctx* ctx)
{
    const int64_t ng = (int64_t)youngs_modulus.size();
    for (int full = 1; full >= 0; full--) {  // "EnergySegmentStrain" is registered first (:11), then the elasticity-only one (:57)
        auto& conn = full ? conn_complete : conn_elasticity_only;
        if (conn.empty()) continue;
        BindList B(ctx, stark);
        for (int k = 0; k < 2; k++) B.add_id(dyn->id_v1, 3, 2 + k);
        for (int k = 0; k < 2; k++) B.add_id(dyn->id_x0, 3, 2 + k);
        for (int k = 0; k < 2; k++) B.add_id(dyn->id_X, 3, 2 + k);
        B.add(scale.data(), ng, 1, 1);
        B.add(section_radius.data(), ng, 1, 1);
        B.add(youngs_modulus.data(), ng, 1, 1);
        if (full) {
            B.add(strain_damping.data(), ng, 1, 1);
            B.add(strain_limit.data(), ng, 1, 1);
            B.add(strain_limit_stiffness.data(), ng, 1, 1);
        }
        B.add_id(stark.dt_array(), 1, -1);
        B.potential(full ? "EnergySegmentStrain" : "EnergySegmentStrain_Elasticity_Only", conn);
    }
}
EnergySegmentStrain::Handler EnergySegmentStrain::add(const PointSetHandler& set, const std::vector<std::array<int, 2>>& segments, const Params& p)
{
    const int group = (int)youngs_modulus.size();
    elasticity_only.push_back(p.elasticity_only);
    scale.push_back(p.scale);
    section_radius.push_back(p.section_radius);
    youngs_modulus.push_back(p.youngs_modulus);
    strain_damping.push_back(p.damping);
    strain_limit.push_back(p.strain_limit);
    strain_limit_stiffness.push_back(p.strain_limit_stiffness);
    auto& conn = p.elasticity_only ? conn_elasticity_only : conn_complete;
    for (const auto& e : segments) {
        const auto g = set.get_global_indices(e);
        conn.push_back({(int32_t)conn.size(), group, g[0], g[1]});
    }
    stark.mark_registration_dirty();
    return Handler(this, group);
}
EnergySegmentStrain::Params EnergySegmentStrain::get_params(const Handler& h) const
{
    const int g = h.idx;
    Params p;
    p.elasticity_only = elasticity_only[g];
    p.scale = scale[g];
    p.section_radius = section_radius[g];
    p.youngs_modulus = youngs_modulus[g];
    p.damping = strain_damping[g];
    p.strain_limit = strain_limit[g];
    p.strain_limit_stiffness = strain_limit_stiffness[g];
    return p;
}
void EnergySegmentStrain::set_params(const Handler& h, const Params& p)
{
    const int g = h.idx;
    if ((bool)elasticity_only[g] != p.elasticity_only) throw std::runtime_error("EnergySegmentStrain::set_params(): elasticity_only cannot be changed");
    scale[g] = p.scale;
    section_radius[g] = p.section_radius;
    youngs_modulus[g] = p.youngs_modulus;
    strain_damping[g] = p.damping;
    strain_limit[g] = p.strain_limit;
    strain_limit_stiffness[g] = p.strain_limit_stiffness;
    stark.mark_registration_dirty();
}

// ======================================================================================================================
// EnergyAttachments  (stark/src/models/interactions/EnergyAttachments.cpp)
// ======================================================================================================================
EnergyAttachments::EnergyAttachments(Stark& s, spPointDynamics d, spRigidBodyDynamics r) : stark(s), dyn(d), rb(r)
{
    stark.add_model(this);
    stark.callbacks->newton->add_is_converged_state_valid([this]() { return _is_converged_state_valid(); });
}
void EnergyAttachments::register_potentials(mistark_ctx* ctx)
{
    for (int t = 0; t < N_TYPES; t++) id_stiffness[t] = -1;
    auto nodes = [&](BindList& B, int first_col, int n) {  // mws.make_vectors(v1, nodes), then (x0, nodes)
        for (int k = 0; k < n; k++) B.add_id(dyn->id_v1, 3, first_col + k);
        for (int k = 0; k < n; k++) B.add_id(dyn->id_x0, 3, first_col + k);
    };
    auto k_dt = [&](BindList& B, int type, int group_col) {
        B.add(stiffness[type].data(), (int64_t)stiffness[type].size(), 1, group_col);
        id_stiffness[type] = B.b.back().array;
        B.add_id(stark.dt_array(), 1, -1);
    };
    if (!conn_p_p.empty()) {  // :17-35
        BindList B(ctx, stark);
        nodes(B, 1, 2);
        k_dt(B, PointPoint, 0);
        B.potential("EnergyAttachments_d_d_p_p", conn_p_p);
    }
    if (!conn_p_e.empty()) {  // :37-60
        BindList B(ctx, stark);
        nodes(B, 2, 3);
        B.add(bary_p_e[0].data(), (int64_t)bary_p_e.size(), 2, 0);
        k_dt(B, PointEdge, 1);
        B.potential("EnergyAttachments_d_d_p_e", conn_p_e);
    }
    if (!conn_p_t.empty()) {  // :62-85
        BindList B(ctx, stark);
        nodes(B, 2, 4);
        B.add(bary_p_t[0].data(), (int64_t)bary_p_t.size(), 3, 0);
        k_dt(B, PointTriangle, 1);
        B.potential("EnergyAttachments_d_d_p_t", conn_p_t);
    }
    if (!conn_e_e.empty()) {  // :87-111
        BindList B(ctx, stark);
        nodes(B, 2, 4);
        B.add(bary_e_e_0[0].data(), (int64_t)bary_e_e_0.size(), 2, 0);
        B.add(bary_e_e_1[0].data(), (int64_t)bary_e_e_1.size(), 2, 0);
        k_dt(B, EdgeEdge, 1);
        B.potential("EnergyAttachments_d_d_e_e", conn_e_e);
    }
    if (!conn_rb_d.empty()) {  // :113-135: k, dt, the point, the local point, then RigidBodyDynamics::get_x1's v1, w1, t0, q0_
        BindList B(ctx, stark);
        k_dt(B, RigidDeformable, 1);
        B.add_id(dyn->id_v1, 3, 3);
        B.add_id(dyn->id_x0, 3, 3);
        B.add(rb_points_loc[0].data(), (int64_t)rb_points_loc.size(), 3, 0);
        B.add_id(rb->id_v1, 3, 2);
        B.add_id(rb->id_w1, 3, 2);
        B.add_id(rb->id_t0, 3, 2);
        B.add_id(rb->id_q0_, 4, 2);
        B.potential("EnergyAttachments_rb_d", conn_rb_d);
    }
}
EnergyAttachments::Handler EnergyAttachments::new_handler(int type, const Params& params, int& group)
{
    group = (int)stiffness[type].size();
    stiffness[type].push_back(params.stiffness);
    tolerance[type].push_back(params.tolerance);
    handlers_map.push_back({type, group});
    stark.mark_registration_dirty();
    return Handler(this, (int)handlers_map.size() - 1);
}
EnergyAttachments::Handler EnergyAttachments::add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points_0, const std::vector<int>& points_1,
                                                  const Params& params)
{
    if (points_0.size() != points_1.size()) throw std::runtime_error("EnergyAttachments::add() found an invalid number of points.");
    int group;
    Handler h = new_handler(PointPoint, params, group);
    for (size_t i = 0; i < points_0.size(); i++) conn_p_p.push_back({group, set_0.get_global_index(points_0[i]), set_1.get_global_index(points_1[i])});
    return h;
}
EnergyAttachments::Handler EnergyAttachments::add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points, const std::vector<std::array<int, 2>>& edges,
                                                  const std::vector<std::array<double, 2>>& bary, const Params& params)
{
    if (edges.size() != points.size() || bary.size() != points.size()) throw std::runtime_error("EnergyAttachments::add() found an invalid input sizes.");
    int group;
    Handler h = new_handler(PointEdge, params, group);
    for (size_t i = 0; i < edges.size(); i++) {
        conn_p_e.push_back({(int32_t)conn_p_e.size(), group, set_0.get_global_index(points[i]), set_1.get_global_index(edges[i][0]), set_1.get_global_index(edges[i][1])});
        bary_p_e.push_back(bary[i]);
    }
    return h;
}
EnergyAttachments::Handler EnergyAttachments::add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points, const std::vector<std::array<int, 3>>& triangles,
                                                  const std::vector<std::array<double, 3>>& bary, const Params& params)
{
    if (triangles.size() != points.size() || bary.size() != points.size()) throw std::runtime_error("EnergyAttachments::add() found an invalid input sizes.");
    int group;
    Handler h = new_handler(PointTriangle, params, group);
    for (size_t i = 0; i < triangles.size(); i++) {
        conn_p_t.push_back({(int32_t)conn_p_t.size(), group, set_0.get_global_index(points[i]), set_1.get_global_index(triangles[i][0]), set_1.get_global_index(triangles[i][1]),
                            set_1.get_global_index(triangles[i][2])});
        bary_p_t.push_back(bary[i]);
    }
    return h;
}
EnergyAttachments::Handler EnergyAttachments::add(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<std::array<int, 2>>& edges_0,
                                                  const std::vector<std::array<int, 2>>& edges_1, const std::vector<std::array<double, 2>>& bary_0,
                                                  const std::vector<std::array<double, 2>>& bary_1, const Params& params)
{
    const size_t n = edges_0.size();
    if (edges_1.size() != n || bary_0.size() != n || bary_1.size() != n) throw std::runtime_error("EnergyAttachments::add() found an invalid input sizes.");
    int group;
    Handler h = new_handler(EdgeEdge, params, group);
    for (size_t i = 0; i < n; i++) {
        conn_e_e.push_back({(int32_t)conn_e_e.size(), group, set_0.get_global_index(edges_0[i][0]), set_0.get_global_index(edges_0[i][1]), set_1.get_global_index(edges_1[i][0]),
                            set_1.get_global_index(edges_1[i][1])});
        bary_e_e_0.push_back(bary_0[i]);
        bary_e_e_1.push_back(bary_1[i]);
    }
    return h;
}
EnergyAttachments::Handler EnergyAttachments::add(const RigidBodyHandler& body, const PointSetHandler& set, const std::vector<Vec3>& loc, const std::vector<int>& set_points, const Params& params)
{
    if (loc.size() != set_points.size()) throw std::runtime_error("EnergyAttachments::add() found an invalid number of points.");
    int group;
    Handler h = new_handler(RigidDeformable, params, group);
    for (size_t i = 0; i < set_points.size(); i++) {
        conn_rb_d.push_back({(int32_t)conn_rb_d.size(), group, body.get_idx(), set.get_global_index(set_points[i])});
        rb_points_loc.push_back(loc[i]);
    }
    return h;
}
EnergyAttachments::Handler EnergyAttachments::add(const RigidBodyHandler& body, const PointSetHandler& set, const std::vector<int>& points, const Params& params)
{
    // :322-333: the points' current positions, expressed in the body's frame
    std::vector<Vec3> loc(points.size());
    for (size_t i = 0; i < points.size(); i++) loc[i] = body.transform_global_to_local_point(dyn->x1[set.get_global_index(points[i])]);
    return add(body, set, loc, points, params);
}
// ---- add_by_distance (EnergyAttachments.cpp:229-297, 334-360): the reference asks tmd::TriangleMeshDistance (extern/TriangleMeshDistance,
// header-only, a sphere-tree over the triangles) for the nearest triangle, its nearest entity and the barycentric coordinates there. The
// answer is a property of the geometry, not of the search structure: here every triangle is tested (scene set-up code, run once), with
// the closest-point regions of Ericson, Real-Time Collision Detection 5.1.5, in the order tmd evaluates them, so that a point on a
// region boundary is classified the same way. Among triangles at exactly the same distance (a shared vertex or edge) either search
// reports the same vertex / the same edge; the edge's two ends may come in the other order, with the weights swapped accordingly.
namespace
{
enum Nearest { N_V0, N_V1, N_V2, N_E01, N_E12, N_E02, N_F };
struct NearestOnTriangle
{
    double dist_sq;
    Nearest entity;
    Vec3 bary, point;
};
NearestOnTriangle nearest_on_triangle(const Vec3& p, const Vec3& a, const Vec3& b, const Vec3& c)
{
    NearestOnTriangle r{};
    auto finish = [&](Nearest e, const Vec3& w) {
        r.entity = e;
        r.bary = w;
        r.point = w[0] * a + w[1] * b + w[2] * c;
        if (e == N_V0) r.point = a;
        if (e == N_V1) r.point = b;
        if (e == N_V2) r.point = c;
        if (e == N_E01) r.point = w[0] * a + w[1] * b;
        if (e == N_E12) r.point = w[1] * b + w[2] * c;
        if (e == N_E02) r.point = w[0] * a + w[2] * c;
        r.dist_sq = dot(p - r.point, p - r.point);
        return r;
    };
    const Vec3 ab = b - a, ac = c - a, bc = c - b;
    // projections of p on the three edge lines, as numerator / denominator pairs (the sign pattern picks the Voronoi region)
    const double s_n = dot(p - a, ab), s_d = dot(p - b, a - b);
    const double t_n = dot(p - a, ac), t_d = dot(p - c, a - c);
    if (s_n <= 0.0 && t_n <= 0.0) return finish(N_V0, {1.0, 0.0, 0.0});
    const double u_n = dot(p - b, bc), u_d = dot(p - c, b - c);
    if (s_d <= 0.0 && u_n <= 0.0) return finish(N_V1, {0.0, 1.0, 0.0});
    if (t_d <= 0.0 && u_d <= 0.0) return finish(N_V2, {0.0, 0.0, 1.0});
    const Vec3 n = cross(ab, ac);
    const double vc = dot(n, cross(a - p, b - p));
    if (vc <= 0.0 && s_n >= 0.0 && s_d >= 0.0) {
        const double w = s_n / (s_n + s_d);
        return finish(N_E01, {1.0 - w, w, 0.0});
    }
    const double va = dot(n, cross(b - p, c - p));
    if (va <= 0.0 && u_n >= 0.0 && u_d >= 0.0) {
        const double w = u_n / (u_n + u_d);
        return finish(N_E12, {0.0, 1.0 - w, w});
    }
    const double vb = dot(n, cross(c - p, a - p));
    if (vb <= 0.0 && t_n >= 0.0 && t_d >= 0.0) {
        const double w = t_n / (t_n + t_d);
        return finish(N_E02, {1.0 - w, 0.0, w});
    }
    const double u = va / (va + vb + vc), v = vb / (va + vb + vc);
    return finish(N_F, {u, v, 1.0 - u - v});
}
struct NearestOnMesh
{
    double distance = std::numeric_limits<double>::max();
    int triangle = -1;
    NearestOnTriangle hit{};
};
NearestOnMesh nearest_on_mesh(const Vec3& p, const std::vector<Vec3>& vertices, const std::vector<std::array<int, 3>>& triangles)
{
    NearestOnMesh best;
    double best_sq = std::numeric_limits<double>::max();
    for (size_t t = 0; t < triangles.size(); t++) {
        const auto& tri = triangles[t];
        const NearestOnTriangle h = nearest_on_triangle(p, vertices[tri[0]], vertices[tri[1]], vertices[tri[2]]);
        if (h.dist_sq < best_sq) {
            best_sq = h.dist_sq;
            best.triangle = (int)t;
            best.hit = h;
        }
    }
    best.distance = std::sqrt(best_sq);
    return best;
}
}  // namespace
EnergyAttachments::MultiHandler EnergyAttachments::add_by_distance(const PointSetHandler& set_0, const PointSetHandler& set_1, const std::vector<int>& points,
                                                                   const std::vector<std::array<int, 3>>& triangles, double distance, const Params& params)
{
    // the mesh is set_1 at its current positions (:235), triangle indices local to set_1
    std::vector<Vec3> verts((size_t)set_1.size());
    for (int i = 0; i < set_1.size(); i++) verts[i] = set_1.get_position(i);
    for (const auto& t : triangles)
        for (int k = 0; k < 3; k++)
            if (t[k] < 0 || t[k] >= set_1.size()) throw std::runtime_error("EnergyAttachments::add_by_distance: triangle vertex outside the point set");
    std::vector<int> pp0, pp1, pe_p, pt_p;
    std::vector<std::array<int, 2>> pe_e;
    std::vector<std::array<double, 2>> pe_b;
    std::vector<std::array<int, 3>> pt_t;
    std::vector<std::array<double, 3>> pt_b;
    for (const int loc : points) {
        const NearestOnMesh d = nearest_on_mesh(set_0.get_position(loc), verts, triangles);
        if (d.triangle < 0 || !(d.distance < distance)) continue;
        const auto& tri = triangles[d.triangle];
        const Vec3& w = d.hit.bary;
        switch (d.hit.entity) {
            case N_V0: pp0.push_back(loc); pp1.push_back(tri[0]); break;
            case N_V1: pp0.push_back(loc); pp1.push_back(tri[1]); break;
            case N_V2: pp0.push_back(loc); pp1.push_back(tri[2]); break;
            case N_E01: pe_p.push_back(loc); pe_e.push_back({tri[0], tri[1]}); pe_b.push_back({w[0], w[1]}); break;
            case N_E12: pe_p.push_back(loc); pe_e.push_back({tri[1], tri[2]}); pe_b.push_back({w[1], w[2]}); break;
            case N_E02: pe_p.push_back(loc); pe_e.push_back({tri[0], tri[2]}); pe_b.push_back({w[0], w[2]}); break;
            case N_F: pt_p.push_back(loc); pt_t.push_back(tri); pt_b.push_back({w[0], w[1], w[2]}); break;
        }
    }
    // three handlers, empty groups included (:291-296)
    return MultiHandler{{add(set_0, set_1, pp0, pp1, params), add(set_0, set_1, pe_p, pe_e, pe_b, params), add(set_0, set_1, pt_p, pt_t, pt_b, params)}};
}
EnergyAttachments::Handler EnergyAttachments::add_by_distance(const RigidBodyHandler& body, const PointSetHandler& set, const std::vector<Vec3>& loc_vertices,
                                                              const std::vector<std::array<int, 3>>& triangles, const std::vector<int>& set_points, double distance, const Params& params)
{
    std::vector<Vec3> glob(loc_vertices.size());
    for (size_t i = 0; i < loc_vertices.size(); i++) glob[i] = body.transform_local_to_global_point(loc_vertices[i]);
    for (const auto& t : triangles)
        for (int k = 0; k < 3; k++)
            if (t[k] < 0 || (size_t)t[k] >= glob.size()) throw std::runtime_error("EnergyAttachments::add_by_distance: triangle vertex outside the vertex list");
    std::vector<int> pts;
    std::vector<Vec3> loc;
    for (const int p : set_points) {
        const NearestOnMesh d = nearest_on_mesh(set.get_position(p), glob, triangles);
        if (d.triangle < 0 || !(d.distance < distance)) continue;
        pts.push_back(p);
        loc.push_back(body.transform_global_to_local_point(d.hit.point));
    }
    return add(body, set, loc, pts, params);
}
EnergyAttachments::Params EnergyAttachments::get_params(const Handler& h) const
{
    const auto [type, group] = handlers_map.at(h.idx);
    return Params{stiffness[type][group], tolerance[type][group]};
}
void EnergyAttachments::set_params(const Handler& h, const Params& p)
{
    const auto [type, group] = handlers_map.at(h.idx);
    stiffness[type][group] = p.stiffness;
    tolerance[type][group] = p.tolerance;
    if (stark.ctx && id_stiffness[type] >= 0) stark.check(mistark_upload(stark.ctx, id_stiffness[type]));
}
bool EnergyAttachments::_is_converged_state_valid()
{
    // :418-520: per attachment type, the first pair further apart than its group's tolerance doubles that group's stiffness and
    // invalidates the step (which is then redone with the stiffer springs)
    bool any_finite = false;
    for (int t = 0; t < N_TYPES; t++)
        for (double tol : tolerance[t]) any_finite = any_finite || tol < std::numeric_limits<double>::max();
    if (!any_finite) return true;
    stark.check(mistark_dofs_to_host_arrays(stark.ctx));
    stark.check(mistark_download(stark.ctx, dyn->id_x0));
    if (!conn_rb_d.empty()) rb->fetch_velocities();
    const double dt = stark.dt;
    auto X = [&](int i) { return dyn->get_x1(i, dt); };
    bool is_valid = true;
    auto check = [&](int type, int group, const Vec3& d) {
        const double tol = tolerance[type][group];
        if (dot(d, d) > tol * tol) {
            is_valid = false;
            stiffness[type][group] *= 2.0;
            stark.check(mistark_upload(stark.ctx, id_stiffness[type]));
            return true;
        }
        return false;
    };
    for (const auto& c : conn_p_p)
        if (check(PointPoint, c[0], X(c[1]) - X(c[2]))) break;
    for (const auto& c : conn_p_e) {
        const auto& b = bary_p_e[c[0]];
        if (check(PointEdge, c[1], X(c[2]) - (b[0] * X(c[3]) + b[1] * X(c[4])))) break;
    }
    for (const auto& c : conn_p_t) {
        const auto& b = bary_p_t[c[0]];
        if (check(PointTriangle, c[1], X(c[2]) - (b[0] * X(c[3]) + b[1] * X(c[4]) + b[2] * X(c[5])))) break;
    }
    for (const auto& c : conn_e_e) {
        const auto &b0 = bary_e_e_0[c[0]], &b1 = bary_e_e_1[c[0]];
        if (check(EdgeEdge, c[1], (b0[0] * X(c[2]) + b0[1] * X(c[3])) - (b1[0] * X(c[4]) + b1[1] * X(c[5])))) break;
    }
    for (const auto& c : conn_rb_d)
        if (check(RigidDeformable, c[1], rb->get_x1(c[2], rb_points_loc[c[0]], dt) - X(c[3]))) break;
    return is_valid;
}

// ======================================================================================================================
// Line presets  (stark/src/models/presets/DeformablesPresets.cpp:11-29, deformables_preset_types.cpp:17-37)
// ======================================================================================================================
Line::Params Line::Params::Elastic_Rubberband()
{
    Params p;
    p.inertia.density = 0.05;
    p.inertia.damping = 0.1;
    p.strain.elasticity_only = false;
    p.strain.section_radius = 0.002;
    p.strain.youngs_modulus = 1e4;
    p.strain.strain_limit = 0.1;
    p.strain.strain_limit_stiffness = 1e5;
    p.strain.damping = 1e-4;
    return p;
}
Line::Handler DeformablesPresets::add_line(const std::string& label, const std::vector<Vec3>& V, const std::vector<std::array<int, 2>>& segments, const Line::Params& p)
{
    PointSetHandler ps = deformables->point_sets->add(V, label);
    auto inertia = deformables->lumped_inertia->add(ps, segments, p.inertia);
    auto strain = deformables->segment_strain->add(ps, segments, p.strain);
    ContactHandler contact = interactions->contact->add_edges(ps, segments, p.contact);
    if (!label.empty() && interactions->output) interactions->output->add_segment_mesh(label, ps, segments);
    return {ps, inertia, strain, contact};
}
Line::VCH DeformablesPresets::add_line_as_segments(const std::string& label, const Vec3& begin, const Vec3& end, int n_segments, const Line::Params& p)
{
    // generate_segment_line (stark/src/utils/mesh_generators.cpp:387-398)
    std::vector<Vec3> V(n_segments + 1);
    std::vector<std::array<int, 2>> S(n_segments);
    for (int i = 0; i <= n_segments; i++) V[i] = begin + (i / (double)n_segments) * (end - begin);
    for (int i = 0; i < n_segments; i++) S[i] = {i, i + 1};
    return {V, S, add_line(label, V, S, p)};
}

}  // namespace mistark
