// host/contact_host.cpp — host half of stark::EnergyFrictionalContact (stark/src/models/interactions/EnergyFrictionalContact.cpp):
// parameters, collision mesh bookkeeping, adaptive contact stiffness and the solver callbacks. Proximity / intersection
// detection and the contact and friction tables live on the device (include/mistark_contact.h).
#include <algorithm>
#include <stdexcept>

#include "sim.hpp"

namespace mistark {

EnergyFrictionalContact::EnergyFrictionalContact(Stark& s, spPointDynamics d, spRigidBodyDynamics r) : stark(s), dyn(d), rb(r)
{
    if (!stark.settings.simulation.init_frictional_contact) return;  // EnergyFrictionalContact.cpp:17
    is_initialized = true;
    stark.add_model(this);
    // :21-28, same order
    stark.callbacks->add_before_time_step([this]() { _before_time_step(); });
    stark.callbacks->newton->add_before_energy_evaluation([this]() { _before_energy_evaluation(); });
    stark.callbacks->newton->add_is_intermediate_state_valid([this]() { return _is_intermediate_state_valid(false); });
    stark.callbacks->newton->add_is_initial_state_valid([this]() { return _is_intermediate_state_valid(true); });
    stark.callbacks->newton->add_is_converged_state_valid([this]() { return _is_intermediate_state_valid(false); });
    stark.callbacks->newton->add_on_intermediate_state_invalid([this]() { _on_intermediate_state_invalid(); });
    stark.callbacks->add_on_time_step_accepted([this]() { _on_time_step_accepted(); });
    stark.callbacks->add_should_continue_execution([this]() { return _should_continue_execution(); });
}
void EnergyFrictionalContact::set_global_params(const GlobalParams& p)
{
    global_params = p;
    contact_stiffness = p.min_contact_stiffness;  // :44-48
    stark.mark_registration_dirty();
}
double EnergyFrictionalContact::_init_contact_thickness(double t) const
{
    // :133-150
    if (!is_initialized) return t;
    if (t == 0.0) {
        if (global_params.default_contact_thickness > 0.0) t = global_params.default_contact_thickness;
        else throw std::runtime_error("Undefined contact thickness found. Explicitly declare per-object contact thickness or set a global default.");
    }
    return t;
}
EnergyFrictionalContact::Handler EnergyFrictionalContact::add_triangles(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles, const Params& params)
{
    return add_triangles(set, triangles, set.all(), params);
}
EnergyFrictionalContact::Handler EnergyFrictionalContact::add_triangles(const PointSetHandler& set, const std::vector<std::array<int, 3>>& triangles,
                                                                        const std::vector<int>& point_set_map, const Params& params)
{
    // :59-65, :173-197
    const int group = (int)contact_thicknesses.size();
    contact_thicknesses.push_back(_init_contact_thickness(params.contact_thickness));
    Mesh m;
    m.kind = MISTARK_CONTACT_DEFORMABLE;
    m.idx_in_ps = set.get_idx();
    for (int l : point_set_map) m.verts.push_back(set.get_global_index(l));
    m.triangles = triangles;
    find_edges_from_simplices(m.edges, triangles, (int)point_set_map.size());
    meshes.push_back(std::move(m));
    stark.mark_registration_dirty();
    return Handler{this, group};
}
EnergyFrictionalContact::Handler EnergyFrictionalContact::add_edges(const PointSetHandler& set, const std::vector<std::array<int, 2>>& edges, const Params& params)
{
    // :66-77, :151-172
    const int group = (int)contact_thicknesses.size();
    contact_thicknesses.push_back(_init_contact_thickness(params.contact_thickness));
    Mesh m;
    m.kind = MISTARK_CONTACT_DEFORMABLE;
    m.idx_in_ps = set.get_idx();
    for (int l = 0; l < set.size(); l++) m.verts.push_back(set.get_global_index(l));
    m.edges = edges;
    meshes.push_back(std::move(m));
    stark.mark_registration_dirty();
    return Handler{this, group};
}
EnergyFrictionalContact::Handler EnergyFrictionalContact::add_triangles(const RigidBodyHandler& body, const std::vector<Vec3>& vertices,
                                                                        const std::vector<std::array<int, 3>>& triangles, const Params& params)
{
    // :78-84, :198-210: local vertices are appended to rigidbody_local_vertices; rigid self collision is disabled
    const int group = (int)contact_thicknesses.size();
    contact_thicknesses.push_back(_init_contact_thickness(params.contact_thickness));
    Mesh m;
    m.kind = MISTARK_CONTACT_RIGIDBODY;
    m.idx_in_ps = body.get_idx();
    for (size_t i = 0; i < vertices.size(); i++) m.verts.push_back((int32_t)(rigidbody_local_vertices.size() + i));
    rigidbody_local_vertices.insert(rigidbody_local_vertices.end(), vertices.begin(), vertices.end());
    m.triangles = triangles;
    find_edges_from_simplices(m.edges, triangles, (int)vertices.size());
    meshes.push_back(std::move(m));
    stark.mark_registration_dirty();
    return Handler{this, group};
}
// (Handler::exit_if_not_valid in the reference: a handle that names no registered collision mesh is an error, here an exception)
static void check_contact_handle(int idx, size_t n_objects, const char* what)
{
    if (idx < 0 || (size_t)idx >= n_objects) throw std::runtime_error(std::string("EnergyFrictionalContact::") + what + ": invalid contact handler");
}
void EnergyFrictionalContact::set_contact_thickness(const Handler& obj, double t)
{
    if (t <= 0.0) throw std::runtime_error("Contact thickness must be positive in EnergyFrictionalContact.");
    check_contact_handle(obj.get_idx(), contact_thicknesses.size(), "set_contact_thickness");
    contact_thicknesses[obj.get_idx()] = t;
    stark.mark_registration_dirty();
}
void EnergyFrictionalContact::set_friction(const Handler& a, const Handler& b, double mu)
{
    check_contact_handle(a.get_idx(), contact_thicknesses.size(), "set_friction");
    check_contact_handle(b.get_idx(), contact_thicknesses.size(), "set_friction");
    if (mu < 0.0) throw std::runtime_error("EnergyFrictionalContact::set_friction: negative friction coefficient");
    friction_pairs.push_back({(double)std::min(a.get_idx(), b.get_idx()), (double)std::max(a.get_idx(), b.get_idx()), mu});
    stark.mark_registration_dirty();
}
void EnergyFrictionalContact::disable_collision(const Handler& a, const Handler& b)
{
    check_contact_handle(a.get_idx(), contact_thicknesses.size(), "disable_collision");
    check_contact_handle(b.get_idx(), contact_thicknesses.size(), "disable_collision");
    disabled_pairs.push_back({std::min(a.get_idx(), b.get_idx()), std::max(a.get_idx(), b.get_idx())});
    stark.mark_registration_dirty();
}
void EnergyFrictionalContact::register_potentials(mistark_ctx* ctx)
{
    id_k = -1;
    k_uploaded = -1.0;
    if (!is_initialized || meshes.empty()) return;
    mistark_contact_arrays a;
    a.v1 = dyn->id_v1;
    a.x0 = dyn->id_x0;
    a.X = dyn->id_X;
    a.dt = stark.dt_array();
    stark.check(id_k = a.k = mistark_array(ctx, &contact_stiffness, 1, 1));
    stark.check(a.thickness = mistark_array(ctx, contact_thicknesses.data(), (int64_t)contact_thicknesses.size(), 1));
    stark.check(a.epsv = mistark_array(ctx, &global_params.friction_stick_slide_threshold, 1, 1));
    a.rb_xloc = -1;
    if (!rigidbody_local_vertices.empty()) stark.check(a.rb_xloc = mistark_array(ctx, rigidbody_local_vertices[0].data(), (int64_t)rigidbody_local_vertices.size(), 3));
    a.rb_v1 = rb->id_v1;
    a.rb_w1 = rb->id_w1;
    a.rb_t0 = rb->id_t0;
    a.rb_q0 = rb->id_q0_;
    stark.check(mistark_contact_init(ctx, &a));
    for (const Mesh& m : meshes) {
        stark.check(mistark_contact_add_mesh(ctx, m.kind, m.idx_in_ps, m.verts.data(), (int32_t)m.verts.size(), m.triangles.empty() ? nullptr : m.triangles[0].data(),
                                             (int32_t)m.triangles.size(), m.edges.empty() ? nullptr : m.edges[0].data(), (int32_t)m.edges.size()));
    }
    for (const auto& f : friction_pairs) stark.check(mistark_contact_set_friction(ctx, (int)f[0], (int)f[1], f[2]));
    for (const auto& d : disabled_pairs) stark.check(mistark_contact_disable_collision(ctx, d[0], d[1]));
    stark.check(mistark_contact_enable(ctx, global_params.triangle_point_enabled, global_params.edge_edge_enabled));
    k_uploaded = contact_stiffness;
}
void EnergyFrictionalContact::_sync_stiffness()
{
    if (id_k >= 0 && contact_stiffness != k_uploaded) {
        stark.check(mistark_upload(stark.ctx, id_k));
        k_uploaded = contact_stiffness;
    }
}
void EnergyFrictionalContact::_before_time_step()
{
    // :531-543: lagged friction from the configuration at the beginning of the step
    if (!global_params.collisions_enabled || !global_params.friction_enabled || is_empty()) return;
    _sync_stiffness();
    stark.check(mistark_contact_update_friction(stark.ctx, &last_n_friction_contacts));
}
void EnergyFrictionalContact::_before_energy_evaluation()
{
    // :368-379
    if (!global_params.collisions_enabled || is_empty()) return;
    _sync_stiffness();
    stark.check(mistark_contact_update(stark.ctx, stark.dt, &last_n_contacts));
    n_detections++;
}
bool EnergyFrictionalContact::_is_intermediate_state_valid(bool is_initial_check)
{
    // :774-799
    if (!global_params.collisions_enabled || !global_params.intersection_test_enabled || is_empty()) return true;
    int64_t n = 0;
    stark.check(mistark_contact_count_intersections(stark.ctx, stark.dt, &n));
    if (n > 0 && is_initial_check && stark.settings.output.enable_output) std::printf("Stark error: Initial collision detected (%lld intersecting edge-triangle pairs)\n", (long long)n);
    return n == 0;
}
void EnergyFrictionalContact::_on_intermediate_state_invalid()
{
    // :800-806
    contact_stiffness *= 2.0;
    _sync_stiffness();
    if (stark.settings.output.enable_output) std::printf("Penetration couldn't be avoided. Contact stiffness hardened to %.1e.\n", contact_stiffness);
}
void EnergyFrictionalContact::_on_time_step_accepted()
{
    contact_stiffness = std::max(global_params.min_contact_stiffness, 0.99 * contact_stiffness);  // :807-810
    _sync_stiffness();
}
bool EnergyFrictionalContact::_should_continue_execution()
{
    // :812-823
    if (!global_params.collisions_enabled) return true;
    return !(contact_stiffness > global_params.max_contact_stiffness);
}

}  // namespace mistark
