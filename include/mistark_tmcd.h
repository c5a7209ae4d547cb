/* mistark_tmcd.h — the collision detector of the contact pipeline as a standalone service: broad phase + narrow phase on the MI355X for
 * callers that hold vertex positions on the host, i.e. the interface a replacement of the reference's collision-detection dependency binds.
 *
 * The reference reaches its detector (TriangleMeshCollisionDetection, "tmcd", stark/extern/TriangleMeshCollisionDetection) through two
 * classes: tmcd::ProximityDetection (src/ProximityDetection.h:9-52) and tmcd::IntersectionDetection (src/IntersectionDetection.h:9-44);
 * stark/src/models/interactions/EnergyFrictionalContact.cpp:116-117,168-169,193-194 (meshes, blacklists), :251-269 (runs), :384-530,
 * :592-770, :781-790 (what it reads from the results). One mistark_cd object stands for one such C++ object; shim/include/
 * TriangleMeshCollisionDetection wraps it in those two classes so that the UNMODIFIED EnergyFrictionalContact.cpp compiles against it
 * (INTEGRATION.md section 6).
 *
 * Results are the reference's six proximity lists and its edge-triangle intersection list with the same members, as flat int32 rows
 * (set = mesh id as returned by add_mesh, idx / vertices local to the mesh):
 *   list 0  point_triangle.point_point     p.set p.idx | T.set T.idx T.v0 T.v1 T.v2 | closest triangle vertex          (8 columns)
 *   list 1  point_triangle.point_edge      p.set p.idx | T.set T.idx T.v0 T.v1 T.v2 | closest triangle edge v0 v1      (9)
 *   list 2  point_triangle.point_triangle  p.set p.idx | T.set T.idx T.v0 T.v1 T.v2                                    (7)
 *   list 3  edge_edge.point_point          first: E.set E.idx E.v0 E.v1 point | second: E.set E.idx E.v0 E.v1 point    (10)
 *   list 4  edge_edge.point_edge           first: E.set E.idx E.v0 E.v1 point | second: E.set E.idx E.v0 E.v1          (9)
 *   list 5  edge_edge.edge_edge            first: E.set E.idx E.v0 E.v1       | second: E.set E.idx E.v0 E.v1          (8)
 *   intersections                          edge: set idx v0 v1 | triangle: set idx v0 v1 v2                            (9)
 * with the distance of every proximity pair (ProximityPair::distance). Pair SETS equal the reference's (same boxes: float, rounded outwards,
 * enlarged, AABBs.cpp:15-45; same closest-feature decisions, ipc_toolkit_geometry_functions.cpp:38-330); the order inside a list is
 * the detector's sorted key order instead of the reference's thread-merge order. All functions return 0 or a negative error code
 * (mistark_cd_last_error). */
#ifndef MISTARK_TMCD_H
#define MISTARK_TMCD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mistark_cd mistark_cd;

int mistark_cd_create(mistark_cd** out, int device);
void mistark_cd_destroy(mistark_cd* cd);
const char* mistark_cd_last_error(mistark_cd* cd);
/* ProximityDetection::add_mesh / IntersectionDetection::add_mesh: `xm` (3 doubles per vertex) is KEPT and read again at every run, as the
 * reference's detector does; connectivity is copied. Returns the mesh id (>= 0). */
int mistark_cd_add_mesh(mistark_cd* cd, const double* xm, int32_t n_vertices, const int32_t* triangles, int32_t n_triangles, const int32_t* edges, int32_t n_edges);
/* add_blacklist(mesh a, mesh b): no pairs between the two meshes (a == b: none inside the mesh) */
int mistark_cd_add_blacklist(mistark_cd* cd, int32_t mesh_a, int32_t mesh_b);
/* add_blacklist_range_point_triangle (edge_edge = 0: points [a0, a1) of mesh_a never pair with triangles [b0, b1) of mesh_b) and
 * add_blacklist_range_edge_edge (edge_edge = 1: edges [a0, a1) of mesh_a with edges [b0, b1) of mesh_b; the first interval must be the lower one in
 * the detector's global edge numbering = registration order, as the reference demands): ProximityDetection.h:24-25, half-open local intervals. */
int mistark_cd_add_blacklist_range(mistark_cd* cd, int32_t edge_edge, int32_t mesh_a, int32_t a0, int32_t a1, int32_t mesh_b, int32_t b0, int32_t b1);
/* activate_point_triangle / activate_edge_edge */
int mistark_cd_activate(mistark_cd* cd, int point_triangle, int edge_edge);
/* ProximityDetection::run(enlargement): counts[l] = rows of list l. (Edge pairs whose |ea x eb|^2 is at most 1e-30 never reach a list:
 * the cutoff the reference passes, EnergyFrictionalContact.h:71.) */
int mistark_cd_run_proximity(mistark_cd* cd, double enlargement, int32_t counts[6]);
/* rows (counts[list] x columns of the list, see above) and distances of the last run; either pointer may be NULL */
int mistark_cd_get_proximity(mistark_cd* cd, int list, int32_t* rows, double* distance);
/* ProximityDetection::get_broad_phase_results() (ProximityDetection.h:31, types.h:24-33): the candidate pairs of a run at the CURRENT positions —
 * every (point, triangle) and (edge a < edge b) pair whose boxes as the reference builds them (floats to nearest, enlarged by (float)enlargement +
 * eps, AABBs.cpp:7-45) overlap, minus points of their own triangle, edges sharing a vertex, blacklisted meshes and ranges
 * (BroadPhasePTEEBase.cpp:176-262). counts[0] = point-triangle pairs, counts[1] = edge-edge pairs; rows of 4: first.set first.idx second.set
 * second.idx, sorted. The proximity lists of the previous run are invalidated (the next mistark_cd_run_proximity searches again). */
int mistark_cd_run_broad_phase(mistark_cd* cd, double enlargement, int32_t counts[2]);
int mistark_cd_get_broad_phase(mistark_cd* cd, int list, int32_t* rows);
/* IntersectionDetection::run(): number of intersecting (edge, triangle) pairs; their rows with mistark_cd_get_intersections */
int mistark_cd_run_intersection(mistark_cd* cd, int32_t* n_pairs);
int mistark_cd_get_intersections(mistark_cd* cd, int32_t* rows);
#ifdef __cplusplus
}
#endif
#endif
