/* mistark_sim.h — C facade over the C++ host mirror (stark_amd/csrc/host/sim.hpp) of the reference's scene API
 * (stark::Simulation + presets, stark/src/models/Simulation.h:13-42, presets/DeformablesPresets.h). It plays the role
 * pystark's nanobind module plays for the reference (pystark/cpp/models/pystark_Simulation.cpp:5-23): a language binding
 * can drive scenes without touching C++. Return values: 0 / >=0 = ok, < 0 = error (mistark_sim_last_error). */
#ifndef MISTARK_SIM_H
#define MISTARK_SIM_H
#include "mistark.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mistark_sim mistark_sim;

typedef struct mistark_sim_settings
{
    double gravity[3];
    double max_time_step_size;
    int32_t use_adaptive_time_step;
    double time_step_size_success_multiplier;
    double time_step_size_lower_bound;
    int32_t device;
    int32_t mirror_state_to_host;
    int32_t enable_output;
    int32_t init_frictional_contact; /* Settings::Simulation::init_frictional_contact */
    mistark_newton_settings newton;
    /* Settings::Output (stark/src/core/Settings.h:12-24): VTK frames <output_directory>/<simulation_name>_<label>_<frame>.vtk of every
     * labelled object at `fps` (< 0: every accepted step); off by default */
    int32_t enable_frame_writes;
    int32_t fps;
    char output_directory[256];
    char simulation_name[64];
    /* Settings::Execution (stark/src/core/Settings.h:33-38): stop conditions of mistark_sim_run (Stark.cpp:84-116) */
    double allowed_execution_time;
    double end_simulation_time;
    int32_t end_frame;
} mistark_sim_settings;
void mistark_sim_default_settings(mistark_sim_settings* s);

/* EnergyLumpedInertia::Params + EnergyTetStrain::Params (stark::Volume::Params) */
typedef struct mistark_volume_params
{
    double density, inertia_damping;
    int32_t quasistatic;
    int32_t elasticity_only;
    double scale, youngs_modulus, poissons_ratio, strain_damping, strain_limit, strain_limit_stiffness;
} mistark_volume_params;
void mistark_volume_params_soft_rubber(mistark_volume_params* p);

/* EnergyLumpedInertia::Params + EnergyTriangleStrain::Params + EnergyDiscreteShells::Params (stark::Surface::Params) */
typedef struct mistark_surface_params
{
    double density, inertia_damping;
    int32_t quasistatic;
    int32_t elasticity_only;
    double scale, thickness, youngs_modulus, poissons_ratio, strain_damping, strain_limit, strain_limit_stiffness, inflation;
    double bending_stiffness, bending_damping;
    int32_t flat_rest_angle;
} mistark_surface_params;
void mistark_surface_params_cotton_fabric(mistark_surface_params* p);

int mistark_sim_create(const mistark_sim_settings* settings, mistark_sim** out);
void mistark_sim_destroy(mistark_sim* sim);
const char* mistark_sim_last_error(mistark_sim* sim);

/* presets->deformables->add_volume_grid / add_volume / add_surface_grid / add_surface: return the point-set index */
int mistark_sim_add_volume_grid(mistark_sim* sim, const char* label, const double center[3], const double dim[3], const int32_t subdivisions[3], const mistark_volume_params* p);
int mistark_sim_add_volume(mistark_sim* sim, const char* label, const double* vertices, int64_t n_vertices, const int32_t* tets, int64_t n_tets, const mistark_volume_params* p);
int mistark_sim_add_surface_grid(mistark_sim* sim, const char* label, const double dim[2], const int32_t subdivisions[2], const mistark_surface_params* p);
int mistark_sim_add_surface(mistark_sim* sim, const char* label, const double* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles, const mistark_surface_params* p);
/* deformables->prescribed_positions->add_inside_aabb: returns the group index */
int mistark_sim_prescribe_inside_aabb(mistark_sim* sim, int point_set, const double center[3], const double dim[3], double stiffness, double tolerance);

/* stark::generate_triangle_grid (stark/src/utils/mesh_generators.cpp:100-166): (n0+1)(n1+1) vertices (3 doubles, z = 0) and 2 n0 n1 triangles.
 * Pass NULL outputs to query the counts. */
int mistark_generate_triangle_grid(const double center[2], const double dim[2], const int32_t subdivisions[2], double* vertices, int64_t* n_vertices, int32_t* triangles,
                                   int64_t* n_triangles);
/* stark::find_edges_from_simplices for triangles (stark/src/utils/mesh_utils.cpp): unique edges in the reference's order. edges may be NULL (count only). */
int mistark_find_edges_from_triangles(const int32_t* triangles, int64_t n_triangles, int64_t n_vertices, int32_t* edges, int64_t* n_edges);
/* deformables->prescribed_positions->add_outside_aabb (EnergyPrescribedPositions.cpp:66-78) */
int mistark_sim_prescribe_outside_aabb(mistark_sim* sim, int point_set, const double center[3], const double dim[3], double stiffness, double tolerance);
/* deformables->prescribed_positions->add(set, points, params): returns the group index */
int mistark_sim_prescribe_points(mistark_sim* sim, int point_set, const int32_t* points, int64_t n, double stiffness, double tolerance);

/* ---- rods (stark::Line presets, DeformablesPresets.cpp:11-29; EnergySegmentStrain::Params) ---------------------------------- */
typedef struct mistark_line_params
{
    double density, inertia_damping;
    int32_t quasistatic;
    int32_t elasticity_only;
    double scale, section_radius, youngs_modulus, strain_damping, strain_limit, strain_limit_stiffness;
} mistark_line_params;
void mistark_line_params_elastic_rubberband(mistark_line_params* p);
/* presets->deformables->add_line / add_line_as_segments: return the point-set index */
int mistark_sim_add_line(mistark_sim* sim, const char* label, const double* vertices, int64_t n_vertices, const int32_t* segments, int64_t n_segments, const mistark_line_params* p);
int mistark_sim_add_line_as_segments(mistark_sim* sim, const char* label, const double begin[3], const double end[3], int32_t n_segments, const mistark_line_params* p);

/* ---- attachments (stark::EnergyAttachments::add overloads, EnergyAttachments.cpp:138-333); point / edge / triangle indices are
 * local to their point set, bary are 2 or 3 doubles per attachment; tolerance <= 0 means none. Return the handler index. ---- */
int mistark_sim_attach_point_point(mistark_sim* sim, int set_0, int set_1, const int32_t* points_0, const int32_t* points_1, int64_t n, double stiffness, double tolerance);
int mistark_sim_attach_point_edge(mistark_sim* sim, int set_0, int set_1, const int32_t* points, const int32_t* edges, const double* bary, int64_t n, double stiffness, double tolerance);
int mistark_sim_attach_point_triangle(mistark_sim* sim, int set_0, int set_1, const int32_t* points, const int32_t* triangles, const double* bary, int64_t n, double stiffness,
                                      double tolerance);
int mistark_sim_attach_edge_edge(mistark_sim* sim, int set_0, int set_1, const int32_t* edges_0, const int32_t* edges_1, const double* bary_0, const double* bary_1, int64_t n,
                                 double stiffness, double tolerance);
/* rb_points_loc == NULL: the points' current positions in the body frame (EnergyAttachments.cpp:322-333) */
int mistark_sim_attach_rigid_body(mistark_sim* sim, int rb, int point_set, const double* rb_points_loc, const int32_t* points, int64_t n, double stiffness, double tolerance);
/* EnergyAttachments::get_params(handler).stiffness (doubled by the tolerance check) */
/* EnergyAttachments::add_by_distance (EnergyAttachments.cpp:229-297): every point of `points` (local to set_0) closer than `distance` to the
 * triangle mesh over set_1 (triangle indices local to set_1, current positions) is attached at the nearest vertex, edge or face.
 * handlers_out: the point-point, point-edge and point-triangle groups. */
int mistark_sim_attach_by_distance(mistark_sim* sim, int set_0, int set_1, const int32_t* points, int64_t n_points, const int32_t* triangles, int64_t n_triangles, double distance,
                                   double stiffness, double tolerance, int32_t handlers_out[3]);
/* (:334-360) the same against a triangle mesh given in a rigid body's local frame; the attachment points on the body are the nearest
 * points of the mesh. Returns the handler index. */
int mistark_sim_attach_rigid_body_by_distance(mistark_sim* sim, int rb, int point_set, const double* loc_vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles,
                                              const int32_t* points, int64_t n_points, double distance, double stiffness, double tolerance);
int mistark_sim_attachment_stiffness(mistark_sim* sim, int handler, double* stiffness);

/* PointSetHandler::add_displacement / add_rotation (also at rest pose), before the first step */
int mistark_sim_point_set_add_displacement(mistark_sim* sim, int point_set, const double d[3]);
int mistark_sim_point_set_add_rotation(mistark_sim* sim, int point_set, double angle_deg, const double axis[3], const double pivot[3]);

/* ---- rigid bodies (stark::RigidBodies, RigidBodyPresets::add_box, RigidBodyHandler) ---------------------------------------- */
int mistark_sim_add_rigid_box(mistark_sim* sim, const char* label, double mass, const double size[3]); /* returns the body index */
int mistark_sim_rb_set_translation(mistark_sim* sim, int rb, const double t[3]);
int mistark_sim_rb_add_translation(mistark_sim* sim, int rb, const double t[3]);
int mistark_sim_rb_add_rotation(mistark_sim* sim, int rb, double angle_deg, const double axis[3], const double pivot[3]);
int mistark_sim_rb_set_velocity(mistark_sim* sim, int rb, const double v[3], const double w[3]);
int mistark_sim_rb_set_default_constraint_params(mistark_sim* sim, double stiffness, double tolerance_in_m, double tolerance_in_deg);
/* RigidBodies::add_constraint_<type>(a, b, ...): type is the reference's suffix ("fix", "global_point", "global_direction",
 * "point", "point_on_axis", "distance", "distance_limits", "direction", "angle_limit", "spring", "linear_velocity",
 * "angular_velocity", "attachment", "point_with_angle_limit", "hinge", "hinge_with_angle_limit", "slider", "prismatic_slider",
 * "spring_with_limits", "prismatic_press", "motor"); params = the remaining arguments flattened in the reference's order
 * (points / directions as 3 doubles). b is ignored by the single-body types. */
int mistark_sim_rb_add_constraint(mistark_sim* sim, const char* type, int a, int b, const double* params, int n_params);
/* RigidBodies::add(mass, inertia_local) without a collision mesh (RigidBodies.cpp:14-20); inertia row-major 3x3. Returns the body index. */
int mistark_sim_rb_add(mistark_sim* sim, double mass, const double inertia_local[9]);
/* stark::inertia_tensor_box (stark/src/utils/mesh_utils.h) */
void mistark_inertia_tensor_box(double mass, const double size[3], double out[9]);
/* RigidBodyHandler::add_force_at_centroid / add_torque (RigidBodyHandler.cpp:109-127), global coordinates */
int mistark_sim_rb_add_force_at_centroid(mistark_sim* sim, int rb, const double f[3]);
int mistark_sim_rb_add_torque(mistark_sim* sim, int rb, const double t[3]);
/* number of base constraints of a kind so far ("global_point", "global_direction", "point", "point_on_axis", "distance",
 * "distance_limits", "direction", "angle_limit", "spring", "linear_velocity", "angular_velocity"): the index the next one will get.
 * mistark_sim_rb_add_constraint returns the index of the (last) base constraint it created. */
int mistark_sim_rb_constraint_count(mistark_sim* sim, const char* base_type);
/* RBCFixHandler::set_transformation (rigidbody_constraints_ui.h:369-379): moves the target of a "fix": anchor_point = index of its
 * "global_point", z_lock / x_lock = indices of its two "global_direction" constraints (the counts before the fix was added: g, d, d + 1);
 * rotation row-major 3x3. The README's spinning box calls this from its per-step script. */
int mistark_sim_rb_fix_set_transformation(mistark_sim* sim, int anchor_point, int z_lock, int x_lock, const double translation[3], const double rotation[9]);
/* The constraint handlers' measurements (rigidbody_constraints_ui.h): out = {violation, force or torque} of base constraint `idx`:
 * get_violation_in_m_and_force / get_violation_in_deg_and_torque / get_signed_violation_in_m_and_force /
 * get_signed_spring_displacement_in_m_and_force (which = 0) / get_signed_damper_velocity_and_force (which = 1) /
 * get_signed_velocity_violation_and_force / get_signed_angular_velocity_violation_in_deg_per_s_and_torque; tolerance = the handler's
 * get_tolerance_in_m / get_tolerance_in_deg (nullable). */
int mistark_sim_rb_constraint_measure(mistark_sim* sim, const char* base_type, int idx, int which, double out[2], double* tolerance);
/* Simulation::run(duration) (Simulation.cpp:62-71): steps until the simulated time advanced by `duration`; returns 1 if the last step succeeded */
int mistark_sim_run(mistark_sim* sim, double duration);
/* t1 [3], q1 [4: w x y z], v1 [3], w1 [3] (nullable outputs) */
int mistark_sim_rb_get_state(mistark_sim* sim, int rb, double* t, double* q, double* v, double* w);

/* ---- frictional contact (stark::EnergyFrictionalContact) -------------------------------------------------------------------- */
typedef struct mistark_contact_global_params
{
    double default_contact_thickness, min_contact_stiffness, max_contact_stiffness, friction_stick_slide_threshold;
    int32_t collisions_enabled, friction_enabled, triangle_point_enabled, edge_edge_enabled, intersection_test_enabled;
} mistark_contact_global_params;
void mistark_contact_default_global_params(mistark_contact_global_params* p);
int mistark_sim_set_contact_global_params(mistark_sim* sim, const mistark_contact_global_params* p);
/* contact group (EnergyFrictionalContact::Handler index) of a point set (kind 0) or rigid body (kind 1) added through a preset */
int mistark_sim_contact_group(mistark_sim* sim, int kind, int idx);
int mistark_sim_set_friction(mistark_sim* sim, int group_a, int group_b, double mu);
int mistark_sim_disable_collision(mistark_sim* sim, int group_a, int group_b);
int mistark_sim_get_contact_info(mistark_sim* sim, double* contact_stiffness, int64_t* n_contacts, int64_t* n_friction_contacts, int64_t* n_detections);

/* multi-GPU sharding (mistark.h): every rank builds the same scene, then one of these before the first step */
int mistark_sim_set_dist_rccl(mistark_sim* sim, int rank, int world, const char unique_id[128]);
int mistark_sim_set_dist_local(mistark_sim* sim, mistark_local_group* group, int rank, int world);
int mistark_sim_set_dist_ipc(mistark_sim* sim, mistark_ipc_comm* comm, int rank, int world);  /* a connected communicator (mistark.h "IPC windows") */

/* Replace the Newton settings used by the following steps (stark::core::Settings::newton). */
int mistark_sim_set_newton_settings(mistark_sim* sim, const mistark_newton_settings* s);
/* symx::SolverCallbacks::add_max_allowed_step (solver_utils.h:73; the hook a CCD would use): every registered callback is asked once per line
 * search, the smallest answer < 1 scales the step ([max] stage, NewtonsMethod.cpp:494-506). Runs on the calling thread. */
int mistark_sim_add_max_allowed_step(mistark_sim* sim, double (*f)(void* user), void* user);
/* Stark::run_one_step: 1 = continue, 0 = stop */
int mistark_sim_run_one_step(mistark_sim* sim);

typedef struct mistark_sim_info
{
    double current_time, dt;
    int32_t current_time_step, last_newton_result;
    int64_t n_points, ndofs;
    int64_t total_newton_iterations, total_cg_iterations, total_linear_solves, failed_steps;
    double total_newton_time, total_linear_solve_time;
    double total_eval_pgh_time, total_eval_p_time, total_project_time, total_assembly_time, total_callback_time, total_step_time;
    int64_t total_evaluations;
    mistark_newton_stats last_stats;
} mistark_sim_info;
int mistark_sim_get_info(mistark_sim* sim, mistark_sim_info* info);
/* copies of the PointDynamics arrays (3 doubles per point); which: 0 = X, 1 = x0, 2 = v0, 3 = v1 */
int mistark_sim_get_points(mistark_sim* sim, int which, double* out);
/* writes x0 / v0 (which: 1, 2) from the caller's buffer and uploads the state */
int mistark_sim_set_points(mistark_sim* sim, int which, const double* in);
/* The part of a time step before the Newton solve (Stark.cpp:145-156: before_time_step callbacks = friction tables at the start-of-step
 * geometry, rigid-body caches, v1 = 0) and the Newton callback that precedes every evaluation (contact tables at the current DoFs), as
 * separate calls: what the reference's harness does to take a stage snapshot at a given state (callbacks->run_before_time_step();
 * set_dofs; newton->run_before_energy_evaluation()). */
int mistark_sim_begin_time_step(mistark_sim* sim);
int mistark_sim_before_energy_evaluation(mistark_sim* sim);
/* the engine context (valid after the first step or after mistark_sim_prepare) */
int mistark_sim_prepare(mistark_sim* sim);
mistark_ctx* mistark_sim_engine(mistark_sim* sim);

#ifdef __cplusplus
}
#endif
#endif
