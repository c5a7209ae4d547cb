/* mistark_contact.h — C ABI of the device contact detector (IPC proximity + intersection detection, contact and friction
 * table construction). Drop-in for the part of stark::EnergyFrictionalContact that runs inside the Newton loop:
 *
 *   EnergyFrictionalContact::add_triangles / add_edges           (EnergyFrictionalContact.cpp:54-91)   -> mistark_contact_add_mesh
 *   EnergyFrictionalContact::set_friction / disable_collision    (:102-119)                            -> mistark_contact_set_friction / _disable_collision
 *   _before_energy_evaluation__update_contacts                   (:368-530)                            -> mistark_contact_update
 *   _before_time_step__update_friction_contacts                  (:531-773)                            -> mistark_contact_update_friction
 *   _is_intermediate_state_valid (tmcd::IntersectionDetection)   (:774-799)                            -> mistark_contact_count_intersections
 *   the 21 barrier + 14 friction add_potential calls             (:829-1218)                           -> mistark_contact_init
 *
 * The reference keeps the collision meshes, the tmcd broad/narrow phase and the LabelledConnectivity tables on the host
 * and refills them at every energy evaluation. Here the meshes live in HBM, detection/classification/routing run as HIP
 * kernels, and the tables are written where the potentials' kernels read them; only the row counts cross PCIe.
 * Row CONTENT is the reference's (same columns, same A/B roles); row ORDER is deterministic (sorted by pair id) instead of
 * the reference's thread-dependent order.
 */
#ifndef MISTARK_CONTACT_H
#define MISTARK_CONTACT_H
#include "mistark.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Engine array ids (mistark_array) of the state the contact potentials bind, named after the reference's members.
 * -1 for a physical system the scene does not have. */
typedef struct mistark_contact_arrays
{
    int32_t v1;        /* PointDynamics::v1   (DoF)                       stride 3 */
    int32_t x0;        /* PointDynamics::x0                               stride 3 */
    int32_t X;         /* PointDynamics::X    (rest)                      stride 3 */
    int32_t dt;        /* Stark::dt                                       stride 1, 1 item */
    int32_t k;         /* EnergyFrictionalContact::contact_stiffness      stride 1, 1 item */
    int32_t thickness; /* EnergyFrictionalContact::contact_thicknesses    stride 1, one item per collision mesh (group) */
    int32_t epsv;      /* GlobalParams::friction_stick_slide_threshold    stride 1, 1 item */
    int32_t rb_xloc;   /* EnergyFrictionalContact::rigidbody_local_vertices  stride 3 */
    int32_t rb_v1;     /* RigidBodyDynamics::v1 (DoF)                     stride 3 */
    int32_t rb_w1;     /* RigidBodyDynamics::w1 (DoF)                     stride 3 */
    int32_t rb_t0;     /* RigidBodyDynamics::t0                           stride 3 */
    int32_t rb_q0;     /* RigidBodyDynamics::q0_                          stride 4 */
} mistark_contact_arrays;

enum { MISTARK_CONTACT_DEFORMABLE = 0, MISTARK_CONTACT_RIGIDBODY = 1 };

/* Creates the detector and registers the 35 contact/friction potentials (empty tables) in the dynamic matrix part. */
int mistark_contact_init(mistark_ctx* ctx, const mistark_contact_arrays* arrays);

/* One collision mesh (EnergyFrictionalContact::Handler group). vertex_index[i]: index of collision vertex i in the physical
 * system's vertex array (PointDynamics global index, or rigidbody_local_vertices global index). triangles / edges: local
 * connectivity. Returns the group id. Rigid meshes get their self collision disabled (EnergyFrictionalContact.cpp:208-209). */
int mistark_contact_add_mesh(mistark_ctx* ctx, int kind, int idx_in_ps, const int32_t* vertex_index, int32_t n_vertices, const int32_t* triangles,
                             int32_t n_triangles, const int32_t* edges, int32_t n_edges);
int mistark_contact_set_friction(mistark_ctx* ctx, int group_a, int group_b, double mu);
int mistark_contact_disable_collision(mistark_ctx* ctx, int group_a, int group_b);
int mistark_contact_enable(mistark_ctx* ctx, int point_triangle, int edge_edge);
/* Candidate search: 0 (default) = sweep and prune over sorted boxes, 1 = LDS-tiled all-pairs (ablation / cross-check). Same pair set. */
int mistark_contact_set_broad_phase(mistark_ctx* ctx, int brute_force);

/* Barrier tables for the positions x0 + dt v1 (v1 = the engine's current DoFs). n_contacts: total rows (nullable). */
int mistark_contact_update(mistark_ctx* ctx, double dt, int64_t* n_contacts);
/* Friction tables (connectivity, T, mu, fn, barycentric coordinates) at x0. */
int mistark_contact_update_friction(mistark_ctx* ctx, int64_t* n_contacts);
/* Number of intersecting edge-triangle pairs at x0 + dt v1. */
int mistark_contact_count_intersections(mistark_ctx* ctx, double dt, int64_t* n_found);

/* Parity access: rows of one table (by potential name, e.g. "contact_d_d_pt_pt_cubic"); conn == NULL queries n_rows/stride. */
int mistark_contact_get_table(mistark_ctx* ctx, const char* potential, int32_t* conn, int32_t* n_rows, int32_t* stride);
/* Parity access: friction data of one friction table: T [n x 6], mu [n], fn [n], bary [n x nbary] (nullable outputs). */
int mistark_contact_get_friction_data(mistark_ctx* ctx, const char* potential, double* T, double* mu, double* fn, double* bary, int32_t* nbary);
/* Parity access: collision vertex positions used by the last update [n_collision_vertices x 3]. */
int mistark_contact_get_vertices(mistark_ctx* ctx, double* x, int64_t* n_vertices);
/* Binding recipe of one of the 35 potentials: (role, stride, connectivity column) per binding in the reference's order;
 * roles index the members of mistark_contact_arrays (0..11), 12 = T, 13 = mu, 14 = fn, 15 = bary. Returns n_bindings. */
int mistark_contact_recipe(const char* potential, int32_t* conn_stride, int32_t* roles, int32_t* strides, int32_t* conn_cols);

#ifdef __cplusplus
}
#endif
#endif
