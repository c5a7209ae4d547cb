// host/sim_capi.cpp — extern "C" facade declared in include/mistark_sim.h
#include <cstring>

#include "../../../include/mistark_sim.h"
#include "sim.hpp"

using namespace mistark;

struct mistark_sim
{
    std::unique_ptr<Simulation> sim;
    std::vector<PointSetHandler> sets;
    std::vector<EnergyAttachments::Handler> attachments;
    std::vector<int> set_group;  // contact group of each point set (-1: none)
    std::vector<RigidBodyHandler> bodies;
    std::vector<int> body_group;
    std::string last_error;
};

#define SIM_BEGIN        \
    if (!s) return -1;   \
    int _ret = 0;        \
    (void)_ret;          \
    try {
#define SIM_END                      \
    }                                \
    catch (const std::exception& e)  \
    {                                \
        s->last_error = e.what();    \
        return -1;                   \
    }                                \
    return _ret;

static Volume::Params to_cpp(const mistark_volume_params& p)
{
    Volume::Params q;
    q.inertia.density = p.density;
    q.inertia.damping = p.inertia_damping;
    q.inertia.quasistatic = p.quasistatic != 0;
    q.strain.elasticity_only = p.elasticity_only != 0;
    q.strain.scale = p.scale;
    q.strain.youngs_modulus = p.youngs_modulus;
    q.strain.poissons_ratio = p.poissons_ratio;
    q.strain.damping = p.strain_damping;
    q.strain.strain_limit = p.strain_limit;
    q.strain.strain_limit_stiffness = p.strain_limit_stiffness;
    return q;
}
static Surface::Params to_cpp(const mistark_surface_params& p)
{
    Surface::Params q;
    q.inertia.density = p.density;
    q.inertia.damping = p.inertia_damping;
    q.inertia.quasistatic = p.quasistatic != 0;
    q.strain.elasticity_only = p.elasticity_only != 0;
    q.strain.scale = p.scale;
    q.strain.thickness = p.thickness;
    q.strain.youngs_modulus = p.youngs_modulus;
    q.strain.poissons_ratio = p.poissons_ratio;
    q.strain.damping = p.strain_damping;
    q.strain.strain_limit = p.strain_limit;
    q.strain.strain_limit_stiffness = p.strain_limit_stiffness;
    q.strain.inflation = p.inflation;
    q.bending.scale = p.scale;
    q.bending.stiffness = p.bending_stiffness;
    q.bending.damping = p.bending_damping;
    q.bending.flat_rest_angle = p.flat_rest_angle != 0;
    return q;
}

template <std::size_t N, class T>
static std::vector<std::array<T, N>> rows(const T* p, int64_t n)
{
    std::vector<std::array<T, N>> out((size_t)n);
    for (int64_t i = 0; i < n; i++)
        for (std::size_t k = 0; k < N; k++) out[i][k] = p[N * i + k];
    return out;
}

extern "C" {

void mistark_sim_default_settings(mistark_sim_settings* s)
{
    if (!s) return;
    Settings d;
    for (int i = 0; i < 3; i++) s->gravity[i] = d.simulation.gravity[i];
    s->max_time_step_size = d.simulation.max_time_step_size;
    s->use_adaptive_time_step = d.simulation.use_adaptive_time_step;
    s->time_step_size_success_multiplier = d.simulation.time_step_size_success_multiplier;
    s->time_step_size_lower_bound = d.simulation.time_step_size_lower_bound;
    s->device = 0;
    s->mirror_state_to_host = 1;
    s->enable_output = 0;
    s->init_frictional_contact = 0;  // (the reference defaults to true; scenes of this facade opt in)
    s->newton = d.newton;
    s->enable_frame_writes = 0;
    s->fps = d.output.fps;
    std::memset(s->output_directory, 0, sizeof(s->output_directory));
    std::memset(s->simulation_name, 0, sizeof(s->simulation_name));
    s->allowed_execution_time = d.execution.allowed_execution_time;
    s->end_simulation_time = d.execution.end_simulation_time;
    s->end_frame = d.execution.end_frame;
}
void mistark_volume_params_soft_rubber(mistark_volume_params* p)
{
    if (!p) return;
    const Volume::Params q = Volume::Params::Soft_Rubber();
    p->density = q.inertia.density;
    p->inertia_damping = q.inertia.damping;
    p->quasistatic = q.inertia.quasistatic;
    p->elasticity_only = q.strain.elasticity_only;
    p->scale = q.strain.scale;
    p->youngs_modulus = q.strain.youngs_modulus;
    p->poissons_ratio = q.strain.poissons_ratio;
    p->strain_damping = q.strain.damping;
    p->strain_limit = q.strain.strain_limit;
    p->strain_limit_stiffness = q.strain.strain_limit_stiffness;
}
void mistark_surface_params_cotton_fabric(mistark_surface_params* p)
{
    if (!p) return;
    const Surface::Params q = Surface::Params::Cotton_Fabric();
    p->density = q.inertia.density;
    p->inertia_damping = q.inertia.damping;
    p->quasistatic = q.inertia.quasistatic;
    p->elasticity_only = q.strain.elasticity_only;
    p->scale = q.strain.scale;
    p->thickness = q.strain.thickness;
    p->youngs_modulus = q.strain.youngs_modulus;
    p->poissons_ratio = q.strain.poissons_ratio;
    p->strain_damping = q.strain.damping;
    p->strain_limit = q.strain.strain_limit;
    p->strain_limit_stiffness = q.strain.strain_limit_stiffness;
    p->inflation = q.strain.inflation;
    p->bending_stiffness = q.bending.stiffness;
    p->bending_damping = q.bending.damping;
    p->flat_rest_angle = q.bending.flat_rest_angle;
}

int mistark_sim_create(const mistark_sim_settings* in, mistark_sim** out)
{
    if (!out) return -1;
    *out = nullptr;
    mistark_sim_settings d;
    if (in) d = *in;
    else mistark_sim_default_settings(&d);
    Settings st;
    for (int i = 0; i < 3; i++) st.simulation.gravity[i] = d.gravity[i];
    st.simulation.max_time_step_size = d.max_time_step_size;
    st.simulation.use_adaptive_time_step = d.use_adaptive_time_step != 0;
    st.simulation.time_step_size_success_multiplier = d.time_step_size_success_multiplier;
    st.simulation.time_step_size_lower_bound = d.time_step_size_lower_bound;
    st.execution.device = d.device;
    st.execution.mirror_state_to_host = d.mirror_state_to_host != 0;
    st.output.enable_output = d.enable_output != 0;
    st.simulation.init_frictional_contact = d.init_frictional_contact != 0;
    st.newton = d.newton;
    st.output.enable_frame_writes = d.enable_frame_writes != 0;
    st.output.fps = d.fps;
    st.output.output_directory = std::string(d.output_directory, strnlen(d.output_directory, sizeof(d.output_directory)));
    st.output.simulation_name = std::string(d.simulation_name, strnlen(d.simulation_name, sizeof(d.simulation_name)));
    st.execution.allowed_execution_time = d.allowed_execution_time;
    st.execution.end_simulation_time = d.end_simulation_time;
    st.execution.end_frame = d.end_frame;
    auto* s = new mistark_sim();
    try {
        s->sim = std::make_unique<Simulation>(st);
    } catch (const std::exception&) {
        delete s;
        return -1;
    }
    *out = s;
    return 0;
}
void mistark_sim_destroy(mistark_sim* s) { delete s; }
const char* mistark_sim_last_error(mistark_sim* s) { return s ? s->last_error.c_str() : "null sim"; }

int mistark_sim_add_volume_grid(mistark_sim* s, const char* label, const double center[3], const double dim[3], const int32_t sub[3], const mistark_volume_params* p)
{
    SIM_BEGIN
    std::vector<Vec3> V;
    std::vector<std::array<int, 4>> T;
    generate_tet_grid(V, T, {center[0], center[1], center[2]}, {dim[0], dim[1], dim[2]}, {sub[0], sub[1], sub[2]});
    auto h = s->sim->presets->deformables->add_volume(label ? label : "", V, T, to_cpp(*p));
    s->sets.push_back(h.point_set);
    s->set_group.push_back(h.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
int mistark_sim_add_volume(mistark_sim* s, const char* label, const double* v, int64_t nv, const int32_t* t, int64_t nt, const mistark_volume_params* p)
{
    SIM_BEGIN
    std::vector<Vec3> V((size_t)nv);
    std::vector<std::array<int, 4>> T((size_t)nt);
    for (int64_t i = 0; i < nv; i++) V[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    for (int64_t i = 0; i < nt; i++) T[i] = {t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
    auto h = s->sim->presets->deformables->add_volume(label ? label : "", V, T, to_cpp(*p));
    s->sets.push_back(h.point_set);
    s->set_group.push_back(h.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
int mistark_sim_add_surface_grid(mistark_sim* s, const char* label, const double dim[2], const int32_t sub[2], const mistark_surface_params* p)
{
    SIM_BEGIN
    auto vch = s->sim->presets->deformables->add_surface_grid(label ? label : "", {dim[0], dim[1]}, {sub[0], sub[1]}, to_cpp(*p));
    s->sets.push_back(vch.handler.point_set);
    s->set_group.push_back(vch.handler.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
int mistark_sim_add_surface(mistark_sim* s, const char* label, const double* v, int64_t nv, const int32_t* t, int64_t nt, const mistark_surface_params* p)
{
    SIM_BEGIN
    std::vector<Vec3> V((size_t)nv);
    std::vector<std::array<int, 3>> T((size_t)nt);
    for (int64_t i = 0; i < nv; i++) V[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    for (int64_t i = 0; i < nt; i++) T[i] = {t[3 * i], t[3 * i + 1], t[3 * i + 2]};
    auto h = s->sim->presets->deformables->add_surface(label ? label : "", V, T, to_cpp(*p));
    s->sets.push_back(h.point_set);
    s->set_group.push_back(h.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
int mistark_sim_prescribe_inside_aabb(mistark_sim* s, int ps, const double c[3], const double d[3], double stiffness, double tolerance)
{
    SIM_BEGIN
    if (ps < 0 || ps >= (int)s->sets.size()) throw std::runtime_error("bad point set");
    EnergyPrescribedPositions::Params p;
    p.stiffness = stiffness;
    p.tolerance = tolerance > 0.0 ? tolerance : std::numeric_limits<double>::max();
    auto h = s->sim->deformables->prescribed_positions->add_inside_aabb(s->sets[ps], {c[0], c[1], c[2]}, {d[0], d[1], d[2]}, p);
    _ret = h.get_idx();
    SIM_END
}
int mistark_generate_triangle_grid(const double center[2], const double dim[2], const int32_t sub[2], double* vertices, int64_t* n_vertices, int32_t* triangles, int64_t* n_triangles)
{
    try {
        std::vector<Vec3> V;
        std::vector<std::array<int, 3>> T;
        generate_triangle_grid(V, T, {center[0], center[1]}, {dim[0], dim[1]}, {sub[0], sub[1]});
        if (n_vertices) *n_vertices = (int64_t)V.size();
        if (n_triangles) *n_triangles = (int64_t)T.size();
        if (vertices) std::memcpy(vertices, V.data(), V.size() * sizeof(Vec3));
        if (triangles)
            for (size_t i = 0; i < T.size(); i++)
                for (int k = 0; k < 3; k++) triangles[3 * i + k] = T[i][k];
        return 0;
    } catch (...) {
        return -1;
    }
}
int mistark_find_edges_from_triangles(const int32_t* triangles, int64_t nt, int64_t nv, int32_t* edges, int64_t* n_edges)
{
    try {
        std::vector<std::array<int, 3>> T((size_t)nt);
        for (int64_t i = 0; i < nt; i++) T[i] = {triangles[3 * i], triangles[3 * i + 1], triangles[3 * i + 2]};
        std::vector<std::array<int, 2>> E;
        find_edges_from_simplices(E, T, (int)nv);
        if (n_edges) *n_edges = (int64_t)E.size();
        if (edges)
            for (size_t i = 0; i < E.size(); i++) {
                edges[2 * i] = E[i][0];
                edges[2 * i + 1] = E[i][1];
            }
        return 0;
    } catch (...) {
        return -1;
    }
}
int mistark_sim_prescribe_outside_aabb(mistark_sim* s, int ps, const double c[3], const double d[3], double stiffness, double tolerance)
{
    SIM_BEGIN
    if (ps < 0 || ps >= (int)s->sets.size()) throw std::runtime_error("bad point set");
    EnergyPrescribedPositions::Params p;
    p.stiffness = stiffness;
    p.tolerance = tolerance > 0.0 ? tolerance : std::numeric_limits<double>::max();
    _ret = s->sim->deformables->prescribed_positions->add_outside_aabb(s->sets[ps], {c[0], c[1], c[2]}, {d[0], d[1], d[2]}, p).get_idx();
    SIM_END
}
static Vec3 v3(const double* p) { return {p[0], p[1], p[2]}; }
static PointSetHandler& the_set(mistark_sim* s, int ps)
{
    if (ps < 0 || ps >= (int)s->sets.size()) throw std::runtime_error("bad point set");
    return s->sets[ps];
}
static RigidBodyHandler& the_body(mistark_sim* s, int rb)
{
    if (rb < 0 || rb >= (int)s->bodies.size()) throw std::runtime_error("bad rigid body");
    return s->bodies[rb];
}
int mistark_sim_prescribe_points(mistark_sim* s, int ps, const int32_t* points, int64_t n, double stiffness, double tolerance)
{
    SIM_BEGIN
    EnergyPrescribedPositions::Params p;
    p.stiffness = stiffness;
    p.tolerance = tolerance > 0.0 ? tolerance : std::numeric_limits<double>::max();
    _ret = s->sim->deformables->prescribed_positions->add(the_set(s, ps), std::vector<int>(points, points + n), p).get_idx();
    SIM_END
}
// ---- rods ----------------------------------------------------------------------------------------------------------------
void mistark_line_params_elastic_rubberband(mistark_line_params* o)
{
    const Line::Params p = Line::Params::Elastic_Rubberband();
    *o = mistark_line_params{p.inertia.density, p.inertia.damping, p.inertia.quasistatic ? 1 : 0, p.strain.elasticity_only ? 1 : 0, p.strain.scale, p.strain.section_radius,
                             p.strain.youngs_modulus, p.strain.damping, p.strain.strain_limit, p.strain.strain_limit_stiffness};
}
static Line::Params to_cpp(const mistark_line_params& c)
{
    Line::Params p;
    p.inertia.density = c.density;
    p.inertia.damping = c.inertia_damping;
    p.inertia.quasistatic = c.quasistatic != 0;
    p.strain.elasticity_only = c.elasticity_only != 0;
    p.strain.scale = c.scale;
    p.strain.section_radius = c.section_radius;
    p.strain.youngs_modulus = c.youngs_modulus;
    p.strain.damping = c.strain_damping;
    p.strain.strain_limit = c.strain_limit;
    p.strain.strain_limit_stiffness = c.strain_limit_stiffness;
    return p;
}
int mistark_sim_add_line(mistark_sim* s, const char* label, const double* v, int64_t nv, const int32_t* seg, int64_t ns, const mistark_line_params* p)
{
    SIM_BEGIN
    std::vector<Vec3> V((size_t)nv);
    std::vector<std::array<int, 2>> S((size_t)ns);
    for (int64_t i = 0; i < nv; i++) V[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    for (int64_t i = 0; i < ns; i++) S[i] = {seg[2 * i], seg[2 * i + 1]};
    auto h = s->sim->presets->deformables->add_line(label ? label : "", V, S, to_cpp(*p));
    s->sets.push_back(h.point_set);
    s->set_group.push_back(h.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
int mistark_sim_add_line_as_segments(mistark_sim* s, const char* label, const double begin[3], const double end[3], int32_t n_segments, const mistark_line_params* p)
{
    SIM_BEGIN
    auto vch = s->sim->presets->deformables->add_line_as_segments(label ? label : "", v3(begin), v3(end), n_segments, to_cpp(*p));
    s->sets.push_back(vch.handler.point_set);
    s->set_group.push_back(vch.handler.contact.get_idx());
    _ret = (int)s->sets.size() - 1;
    SIM_END
}
// ---- attachments ---------------------------------------------------------------------------------------------------------
static EnergyAttachments::Params attachment_params(double stiffness, double tolerance)
{
    return EnergyAttachments::Params{stiffness, tolerance > 0.0 ? tolerance : std::numeric_limits<double>::max()};
}
static int keep(mistark_sim* s, const EnergyAttachments::Handler& h)
{
    s->attachments.push_back(h);
    return (int)s->attachments.size() - 1;
}
int mistark_sim_attach_point_point(mistark_sim* s, int s0, int s1, const int32_t* p0, const int32_t* p1, int64_t n, double k, double tol)
{
    SIM_BEGIN
    _ret = keep(s, s->sim->interactions->attachments->add(the_set(s, s0), the_set(s, s1), std::vector<int>(p0, p0 + n), std::vector<int>(p1, p1 + n), attachment_params(k, tol)));
    SIM_END
}
int mistark_sim_attach_point_edge(mistark_sim* s, int s0, int s1, const int32_t* pts, const int32_t* edges, const double* bary, int64_t n, double k, double tol)
{
    SIM_BEGIN
    _ret = keep(s, s->sim->interactions->attachments->add(the_set(s, s0), the_set(s, s1), std::vector<int>(pts, pts + n), rows<2, int>(edges, n), rows<2, double>(bary, n),
                                                          attachment_params(k, tol)));
    SIM_END
}
int mistark_sim_attach_point_triangle(mistark_sim* s, int s0, int s1, const int32_t* pts, const int32_t* tris, const double* bary, int64_t n, double k, double tol)
{
    SIM_BEGIN
    _ret = keep(s, s->sim->interactions->attachments->add(the_set(s, s0), the_set(s, s1), std::vector<int>(pts, pts + n), rows<3, int>(tris, n), rows<3, double>(bary, n),
                                                          attachment_params(k, tol)));
    SIM_END
}
int mistark_sim_attach_edge_edge(mistark_sim* s, int s0, int s1, const int32_t* e0, const int32_t* e1, const double* b0, const double* b1, int64_t n, double k, double tol)
{
    SIM_BEGIN
    _ret = keep(s, s->sim->interactions->attachments->add(the_set(s, s0), the_set(s, s1), rows<2, int>(e0, n), rows<2, int>(e1, n), rows<2, double>(b0, n), rows<2, double>(b1, n),
                                                          attachment_params(k, tol)));
    SIM_END
}
int mistark_sim_attach_rigid_body(mistark_sim* s, int rb, int ps, const double* loc, const int32_t* pts, int64_t n, double k, double tol)
{
    SIM_BEGIN
    const std::vector<int> points(pts, pts + n);
    auto att = s->sim->interactions->attachments;
    if (loc) {
        std::vector<Vec3> L((size_t)n);
        for (int64_t i = 0; i < n; i++) L[i] = v3(loc + 3 * i);
        _ret = keep(s, att->add(the_body(s, rb), the_set(s, ps), L, points, attachment_params(k, tol)));
    } else {
        _ret = keep(s, att->add(the_body(s, rb), the_set(s, ps), points, attachment_params(k, tol)));
    }
    SIM_END
}
int mistark_sim_attach_by_distance(mistark_sim* s, int set_0, int set_1, const int32_t* pts, int64_t n_points, const int32_t* tris, int64_t n_triangles, double distance, double k, double tol,
                                   int32_t handlers_out[3])
{
    SIM_BEGIN
    if (n_points < 0 || n_triangles < 0 || (n_points > 0 && !pts) || (n_triangles > 0 && !tris) || !handlers_out) throw std::runtime_error("attach_by_distance: null array or negative size");
    const std::vector<int> points(pts, pts + n_points);
    std::vector<std::array<int, 3>> T((size_t)n_triangles);
    for (int64_t i = 0; i < n_triangles; i++) T[i] = {tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]};
    auto mh = s->sim->interactions->attachments->add_by_distance(the_set(s, set_0), the_set(s, set_1), points, T, distance, attachment_params(k, tol));
    for (int i = 0; i < 3; i++) handlers_out[i] = keep(s, mh.handlers[i]);
    SIM_END
}
int mistark_sim_attach_rigid_body_by_distance(mistark_sim* s, int rb, int ps, const double* loc_vertices, int64_t n_vertices, const int32_t* tris, int64_t n_triangles, const int32_t* pts,
                                              int64_t n_points, double distance, double k, double tol)
{
    SIM_BEGIN
    if (n_points < 0 || n_triangles < 0 || n_vertices < 0 || (n_points > 0 && !pts) || (n_triangles > 0 && !tris) || (n_vertices > 0 && !loc_vertices))
        throw std::runtime_error("attach_rigid_body_by_distance: null array or negative size");
    const std::vector<int> points(pts, pts + n_points);
    std::vector<Vec3> V((size_t)n_vertices);
    for (int64_t i = 0; i < n_vertices; i++) V[i] = v3(loc_vertices + 3 * i);
    std::vector<std::array<int, 3>> T((size_t)n_triangles);
    for (int64_t i = 0; i < n_triangles; i++) T[i] = {tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]};
    _ret = keep(s, s->sim->interactions->attachments->add_by_distance(the_body(s, rb), the_set(s, ps), V, T, points, distance, attachment_params(k, tol)));
    SIM_END
}
int mistark_sim_attachment_stiffness(mistark_sim* s, int handler, double* stiffness)
{
    SIM_BEGIN
    if (handler < 0 || handler >= (int)s->attachments.size()) throw std::runtime_error("bad attachment handler");
    *stiffness = s->attachments[handler].get_params().stiffness;
    SIM_END
}
int mistark_sim_point_set_add_displacement(mistark_sim* s, int ps, const double d[3])
{
    SIM_BEGIN
    the_set(s, ps).add_displacement(v3(d));
    SIM_END
}
int mistark_sim_point_set_add_rotation(mistark_sim* s, int ps, double angle_deg, const double axis[3], const double pivot[3])
{
    SIM_BEGIN
    the_set(s, ps).add_rotation(angle_deg, v3(axis), pivot ? v3(pivot) : Vec3{0.0, 0.0, 0.0});
    SIM_END
}
int mistark_sim_add_rigid_box(mistark_sim* s, const char* label, double mass, const double size[3])
{
    SIM_BEGIN
    auto vch = s->sim->presets->rigidbodies->add_box(label ? label : "", mass, v3(size));
    s->bodies.push_back(vch.handler.rigidbody);
    s->body_group.push_back(vch.handler.contact.get_idx());
    _ret = (int)s->bodies.size() - 1;
    SIM_END
}
int mistark_sim_rb_set_translation(mistark_sim* s, int rb, const double t[3])
{
    SIM_BEGIN
    the_body(s, rb).set_translation(v3(t));
    SIM_END
}
int mistark_sim_rb_add_translation(mistark_sim* s, int rb, const double t[3])
{
    SIM_BEGIN
    the_body(s, rb).add_translation(v3(t));
    SIM_END
}
int mistark_sim_rb_add_rotation(mistark_sim* s, int rb, double angle_deg, const double axis[3], const double pivot[3])
{
    SIM_BEGIN
    the_body(s, rb).add_rotation(angle_deg, v3(axis), pivot ? v3(pivot) : Vec3{0.0, 0.0, 0.0});
    SIM_END
}
int mistark_sim_rb_set_velocity(mistark_sim* s, int rb, const double v[3], const double w[3])
{
    SIM_BEGIN
    if (v) the_body(s, rb).set_velocity(v3(v));
    if (w) the_body(s, rb).set_angular_velocity(v3(w));
    SIM_END
}
int mistark_sim_rb_set_default_constraint_params(mistark_sim* s, double stiffness, double tol_m, double tol_deg)
{
    SIM_BEGIN
    auto& R = *s->sim->rigidbodies;
    if (stiffness > 0.0) R.set_default_constraint_stiffness(stiffness);
    if (tol_m > 0.0) R.set_default_constraint_distance_tolerance(tol_m);
    if (tol_deg > 0.0) R.set_default_constraint_angle_tolerance(tol_deg);
    SIM_END
}
int mistark_sim_rb_add_constraint(mistark_sim* s, const char* type, int ia, int ib, const double* p, int n)
{
    SIM_BEGIN
    auto& R = *s->sim->rigidbodies;
    const std::string t = type ? type : "";
    RigidBodyHandler& a = the_body(s, ia);
    auto need = [&](int k) {
        if (n != k) throw std::runtime_error("constraint '" + t + "' expects " + std::to_string(k) + " parameters");
    };
    const bool single = t == "fix" || t == "global_point" || t == "global_direction";
    RigidBodyHandler& b = single ? a : the_body(s, ib);
    if (t == "fix") { need(0); R.add_constraint_fix(a); }
    else if (t == "global_point") { need(3); _ret = R.add_constraint_global_point(a, v3(p)); }
    else if (t == "global_direction") { need(3); _ret = R.add_constraint_global_direction(a, v3(p)); }
    else if (t == "point") { need(3); _ret = R.add_constraint_point(a, b, v3(p)); }
    else if (t == "point_on_axis") { need(6); _ret = R.add_constraint_point_on_axis(a, b, v3(p), v3(p + 3)); }
    else if (t == "distance") { need(6); _ret = R.add_constraint_distance(a, b, v3(p), v3(p + 3)); }
    else if (t == "distance_limits") { need(8); _ret = R.add_constraint_distance_limits(a, b, v3(p), v3(p + 3), p[6], p[7]); }
    else if (t == "direction") { need(3); _ret = R.add_constraint_direction(a, b, v3(p)); }
    else if (t == "angle_limit") { need(4); _ret = R.add_constraint_angle_limit(a, b, v3(p), p[3]); }
    else if (t == "spring") { need(8); _ret = R.add_constraint_spring(a, b, v3(p), v3(p + 3), p[6], p[7]); }
    else if (t == "linear_velocity") { need(6); _ret = R.add_constraint_linear_velocity(a, b, v3(p), p[3], p[4], p[5]); }
    else if (t == "angular_velocity") { need(6); _ret = R.add_constraint_angular_velocity(a, b, v3(p), p[3], p[4], p[5]); }
    else if (t == "attachment") { need(0); R.add_constraint_attachment(a, b); }
    else if (t == "point_with_angle_limit") { need(7); R.add_constraint_point_with_angle_limit(a, b, v3(p), v3(p + 3), p[6]); }
    else if (t == "hinge") { need(6); R.add_constraint_hinge(a, b, v3(p), v3(p + 3)); }
    else if (t == "hinge_with_angle_limit") { need(7); R.add_constraint_hinge_with_angle_limit(a, b, v3(p), v3(p + 3), p[6]); }
    else if (t == "slider") { need(6); R.add_constraint_slider(a, b, v3(p), v3(p + 3)); }
    else if (t == "prismatic_slider") { need(6); R.add_constraint_prismatic_slider(a, b, v3(p), v3(p + 3)); }
    else if (t == "spring_with_limits") { need(10); R.add_constraint_spring_with_limits(a, b, v3(p), v3(p + 3), p[6], p[7], p[8], p[9]); }
    else if (t == "prismatic_press") { need(9); R.add_constraint_prismatic_press(a, b, v3(p), v3(p + 3), p[6], p[7], p[8]); }
    else if (t == "motor") { need(9); R.add_constraint_motor(a, b, v3(p), v3(p + 3), p[6], p[7], p[8]); }
    else throw std::runtime_error("unknown rigid body constraint type '" + t + "'");
    SIM_END
}
static EnergyRigidBodyConstraints::Kind base_kind(const std::string& t)
{
    using K = EnergyRigidBodyConstraints;
    static const std::pair<const char*, K::Kind> names[] = {{"global_point", K::GlobalPoints}, {"global_direction", K::GlobalDirections}, {"point", K::Points},
                                                            {"point_on_axis", K::PointOnAxes}, {"distance", K::Distances}, {"distance_limits", K::DistanceLimits},
                                                            {"direction", K::Directions}, {"angle_limit", K::AngleLimits}, {"spring", K::DampedSprings},
                                                            {"linear_velocity", K::LinearVelocity}, {"angular_velocity", K::AngularVelocity}};
    for (const auto& n : names)
        if (t == n.first) return n.second;
    throw std::runtime_error("unknown base rigid body constraint type '" + t + "'");
}
int mistark_sim_rb_add(mistark_sim* s, double mass, const double J[9])
{
    SIM_BEGIN
    Mat3 inertia;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) inertia[3 * i + j] = J[3 * i + j];
    s->bodies.push_back(s->sim->rigidbodies->add(mass, inertia));
    s->body_group.push_back(-1);
    _ret = (int)s->bodies.size() - 1;
    SIM_END
}
void mistark_inertia_tensor_box(double mass, const double size[3], double out[9])
{
    const Mat3 J = inertia_tensor_box(mass, {size[0], size[1], size[2]});
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[3 * i + j] = J[3 * i + j];
}
int mistark_sim_rb_add_force_at_centroid(mistark_sim* s, int rb, const double f[3])
{
    SIM_BEGIN
    the_body(s, rb).add_force_at_centroid(v3(f));
    SIM_END
}
int mistark_sim_rb_add_torque(mistark_sim* s, int rb, const double t[3])
{
    SIM_BEGIN
    the_body(s, rb).add_torque(v3(t));
    SIM_END
}
int mistark_sim_rb_constraint_count(mistark_sim* s, const char* type)
{
    SIM_BEGIN
    _ret = (int)s->sim->rigidbodies->constraints->tables[base_kind(type ? type : "")].conn.size();
    SIM_END
}
int mistark_sim_rb_fix_set_transformation(mistark_sim* s, int anchor_point, int z_lock, int x_lock, const double t[3], const double R[9])
{
    SIM_BEGIN
    Mat3 M;
    for (int k = 0; k < 9; k++) M[k] = R[k];
    s->sim->rigidbodies->set_fix_transformation(anchor_point, z_lock, x_lock, Vec3{t[0], t[1], t[2]}, M);
    SIM_END
}
int mistark_sim_rb_constraint_measure(mistark_sim* s, const char* type, int idx, int which, double out[2], double* tolerance)
{
    SIM_BEGIN
    auto& C = *s->sim->rigidbodies->constraints;
    const auto kind = base_kind(type ? type : "");
    const auto r = C.measure(kind, idx, which);
    out[0] = r[0];
    out[1] = r[1];
    if (tolerance) *tolerance = C.tables[kind].tolerance.at((size_t)idx);
    SIM_END
}
int mistark_sim_run(mistark_sim* s, double duration)
{
    SIM_BEGIN
    _ret = s->sim->get_stark().run(duration) ? 1 : 0;
    SIM_END
}
int mistark_sim_rb_get_state(mistark_sim* s, int rb, double* t, double* q, double* v, double* w)
{
    SIM_BEGIN
    RigidBodyHandler& h = the_body(s, rb);
    if (t) { const Vec3 x = h.get_translation(); std::memcpy(t, x.data(), sizeof(x)); }
    if (q) { const Quat x = h.get_quaternion(); std::memcpy(q, x.data(), sizeof(x)); }
    if (v) { const Vec3 x = h.get_velocity(); std::memcpy(v, x.data(), sizeof(x)); }
    if (w) { const Vec3 x = h.get_angular_velocity(); std::memcpy(w, x.data(), sizeof(x)); }
    SIM_END
}
void mistark_contact_default_global_params(mistark_contact_global_params* p)
{
    if (!p) return;
    const EnergyFrictionalContact::GlobalParams d;
    p->default_contact_thickness = d.default_contact_thickness;
    p->min_contact_stiffness = d.min_contact_stiffness;
    p->max_contact_stiffness = d.max_contact_stiffness;
    p->friction_stick_slide_threshold = d.friction_stick_slide_threshold;
    p->collisions_enabled = d.collisions_enabled;
    p->friction_enabled = d.friction_enabled;
    p->triangle_point_enabled = d.triangle_point_enabled;
    p->edge_edge_enabled = d.edge_edge_enabled;
    p->intersection_test_enabled = d.intersection_test_enabled;
}
int mistark_sim_set_contact_global_params(mistark_sim* s, const mistark_contact_global_params* p)
{
    SIM_BEGIN
    EnergyFrictionalContact::GlobalParams g;
    g.default_contact_thickness = p->default_contact_thickness;
    g.min_contact_stiffness = p->min_contact_stiffness;
    g.max_contact_stiffness = p->max_contact_stiffness;
    g.friction_stick_slide_threshold = p->friction_stick_slide_threshold;
    g.collisions_enabled = p->collisions_enabled != 0;
    g.friction_enabled = p->friction_enabled != 0;
    g.triangle_point_enabled = p->triangle_point_enabled != 0;
    g.edge_edge_enabled = p->edge_edge_enabled != 0;
    g.intersection_test_enabled = p->intersection_test_enabled != 0;
    s->sim->interactions->contact->set_global_params(g);
    SIM_END
}
int mistark_sim_contact_group(mistark_sim* s, int kind, int idx)
{
    SIM_BEGIN
    const std::vector<int>& g = kind == 0 ? s->set_group : s->body_group;
    if (idx < 0 || idx >= (int)g.size()) throw std::runtime_error("bad object index");
    _ret = g[idx];
    if (_ret < 0) throw std::runtime_error("the object has no collision mesh (frictional contact not initialised)");
    SIM_END
}
static ContactHandler group_handler(mistark_sim* s, int g) { return ContactHandler{s->sim->interactions->contact.get(), g}; }
int mistark_sim_set_friction(mistark_sim* s, int a, int b, double mu)
{
    SIM_BEGIN
    s->sim->interactions->contact->set_friction(group_handler(s, a), group_handler(s, b), mu);
    SIM_END
}
int mistark_sim_disable_collision(mistark_sim* s, int a, int b)
{
    SIM_BEGIN
    s->sim->interactions->contact->disable_collision(group_handler(s, a), group_handler(s, b));
    SIM_END
}
int mistark_sim_get_contact_info(mistark_sim* s, double* k, int64_t* n_contacts, int64_t* n_friction, int64_t* n_detections)
{
    SIM_BEGIN
    auto& c = *s->sim->interactions->contact;
    if (k) *k = c.get_contact_stiffness();
    if (n_contacts) *n_contacts = c.last_n_contacts;
    if (n_friction) *n_friction = c.last_n_friction_contacts;
    if (n_detections) *n_detections = c.n_detections;
    SIM_END
}
int mistark_sim_set_dist_rccl(mistark_sim* s, int rank, int world, const char unique_id[128])
{
    SIM_BEGIN
    auto& ex = s->sim->get_stark().settings.execution;
    ex.rank = rank;
    ex.world = world;
    ex.rccl_unique_id.assign(unique_id, unique_id + 128);
    ex.local_group = nullptr;
    ex.ipc_comm = nullptr;
    s->sim->get_stark().mark_registration_dirty();
    SIM_END
}
int mistark_sim_set_dist_local(mistark_sim* s, mistark_local_group* group, int rank, int world)
{
    SIM_BEGIN
    auto& ex = s->sim->get_stark().settings.execution;
    ex.rank = rank;
    ex.world = world;
    ex.local_group = group;
    s->sim->get_stark().mark_registration_dirty();
    SIM_END
}
int mistark_sim_set_dist_ipc(mistark_sim* s, mistark_ipc_comm* comm, int rank, int world)
{
    SIM_BEGIN
    if (!comm) throw std::runtime_error("set_dist_ipc: null communicator");
    auto& ex = s->sim->get_stark().settings.execution;
    ex.rank = rank;
    ex.world = world;
    ex.local_group = nullptr;
    ex.ipc_comm = comm;
    s->sim->get_stark().mark_registration_dirty();
    SIM_END
}
int mistark_sim_set_newton_settings(mistark_sim* s, const mistark_newton_settings* ns)
{
    SIM_BEGIN
    s->sim->get_stark().settings.newton = *ns;
    SIM_END
}
int mistark_sim_add_max_allowed_step(mistark_sim* s, double (*f)(void* user), void* user)
{
    // symx::SolverCallbacks::add_max_allowed_step (solver_utils.h:73): the [max] stage of the line search, NewtonsMethod.cpp:494-506
    SIM_BEGIN
    if (!f) throw std::runtime_error("mistark_sim_add_max_allowed_step: null callback");
    s->sim->get_stark().callbacks->newton->add_max_allowed_step([f, user]() { return f(user); });
    SIM_END
}
int mistark_sim_run_one_step(mistark_sim* s)
{
    SIM_BEGIN
    _ret = s->sim->get_stark().run_one_step() ? 1 : 0;
    SIM_END
}
int mistark_sim_begin_time_step(mistark_sim* s)
{
    SIM_BEGIN
    s->sim->get_stark().begin_time_step();
    SIM_END
}
int mistark_sim_before_energy_evaluation(mistark_sim* s)
{
    SIM_BEGIN
    s->sim->get_stark().before_energy_evaluation();
    SIM_END
}
int mistark_sim_prepare(mistark_sim* s)
{
    SIM_BEGIN
    s->sim->get_stark().ensure_registered();
    SIM_END
}
mistark_ctx* mistark_sim_engine(mistark_sim* s) { return s ? s->sim->get_stark().ctx : nullptr; }

int mistark_sim_get_info(mistark_sim* s, mistark_sim_info* info)
{
    SIM_BEGIN
    Stark& st = s->sim->get_stark();
    info->current_time = st.current_time;
    info->dt = st.dt;
    info->current_time_step = st.current_time_step;
    info->last_newton_result = st.last_newton_result;
    info->n_points = s->sim->deformables->point_sets->size();
    info->ndofs = st.ctx ? mistark_ndofs(st.ctx) : 3 * info->n_points;
    info->total_newton_iterations = st.total_newton_iterations;
    info->total_cg_iterations = st.total_cg_iterations;
    info->total_linear_solves = st.total_linear_solves;
    info->failed_steps = st.failed_steps;
    info->total_newton_time = st.total_newton_time;
    info->total_linear_solve_time = st.total_linear_solve_time;
    info->total_eval_pgh_time = st.total_eval_pgh_time;
    info->total_eval_p_time = st.total_eval_p_time;
    info->total_project_time = st.total_project_time;
    info->total_assembly_time = st.total_assembly_time;
    info->total_callback_time = st.total_callback_time;
    info->total_step_time = st.total_step_time;
    info->total_evaluations = st.total_evaluations;
    info->last_stats = st.last_stats;
    SIM_END
}
int mistark_sim_get_points(mistark_sim* s, int which, double* out)
{
    SIM_BEGIN
    PointDynamics& pd = *s->sim->deformables->point_sets;
    if (which != 0 && s->sim->get_stark().ctx) pd.mirror_to_host();
    const std::vector<Vec3>* src = which == 0 ? &pd.X : which == 1 ? &pd.x0 : which == 2 ? &pd.v0 : which == 3 ? &pd.v1 : nullptr;
    if (!src) throw std::runtime_error("bad array selector");
    if (!src->empty()) std::memcpy(out, (*src)[0].data(), src->size() * sizeof(Vec3));
    SIM_END
}
int mistark_sim_set_points(mistark_sim* s, int which, const double* in)
{
    SIM_BEGIN
    PointDynamics& pd = *s->sim->deformables->point_sets;
    std::vector<Vec3>* dst = which == 1 ? &pd.x0 : which == 2 ? &pd.v0 : nullptr;
    if (!dst) throw std::runtime_error("bad array selector");
    if (s->sim->get_stark().ctx) pd.mirror_to_host();
    if (!dst->empty()) std::memcpy((*dst)[0].data(), in, dst->size() * sizeof(Vec3));
    if (which == 1) pd.x1 = pd.x0;
    if (s->sim->get_stark().ctx) pd.upload_state();
    SIM_END
}

}  // extern "C"
