// host/mesh.hpp — synthetic mesh generators and topology helpers used by the scene layer.
// Restates (same vertex numbering, same element order) the reference's
//   stark::generate_triangle_grid   stark/src/utils/mesh_generators.cpp:100-166
//   stark::generate_tet_grid        stark/src/utils/mesh_generators.cpp:264-380   (12 tets per hexahedron + centre node)
//   stark::find_edges_from_simplices stark/src/utils/mesh_utils.h:153-166
//   stark::find_internal_angles     stark/src/utils/mesh_utils.cpp:217-251
//   stark::triangle_area / unsigned_tetra_volume   stark/src/utils/mesh_utils.cpp:189-200
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iterator>
#include <map>
#include <stdexcept>
#include <vector>

namespace mistark {

using Vec3 = std::array<double, 3>;  // layout-compatible with Eigen::Vector3d (PointDynamics AoS layout)

inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 operator*(double s, const Vec3& a) { return {s * a[0], s * a[1], s * a[2]}; }
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(const Vec3& a, const Vec3& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }

inline double triangle_area(const Vec3& p0, const Vec3& p1, const Vec3& p2) { return 0.5 * norm(cross(p0 - p2, p1 - p2)); }
inline double unsigned_tetra_volume(const Vec3& p0, const Vec3& p1, const Vec3& p2, const Vec3& p3)
{
    return std::abs((1.0 / 6.0) * dot(cross(p1 - p0, p2 - p0), p3 - p0));
}

inline void generate_triangle_grid(std::vector<Vec3>& V, std::vector<std::array<int, 3>>& T, const std::array<double, 2>& center, const std::array<double, 2>& dim,
                                   const std::array<int, 2>& nq, double z = 0.0)
{
    const double bx = center[0] - 0.5 * dim[0], by = center[1] - 0.5 * dim[1];
    const double tx = center[0] + 0.5 * dim[0], ty = center[1] + 0.5 * dim[1];
    const int nx = nq[0] + 1, ny = nq[1] + 1;
    const double dx = (tx - bx) / (double)nq[0], dy = (ty - by) / (double)nq[1];
    V.resize((size_t)nx * ny);
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) V[(size_t)ny * i + j] = {bx + i * dx, by + j * dy, z};
    T.clear();
    T.reserve((size_t)2 * nq[0] * nq[1]);
    for (int ei = 0; ei < nq[0]; ei++)
        for (int ej = 0; ej < nq[1]; ej++) {
            const int n0 = ny * (ei + 0) + (ej + 0), n1 = ny * (ei + 0) + (ej + 1), n2 = ny * (ei + 1) + (ej + 0), n3 = ny * (ei + 1) + (ej + 1);
            if (ei % 2 == ej % 2) {
                T.push_back({n0, n2, n3});
                T.push_back({n0, n3, n1});
            } else {
                T.push_back({n0, n2, n1});
                T.push_back({n2, n3, n1});
            }
        }
}

inline void generate_tet_grid(std::vector<Vec3>& V, std::vector<std::array<int, 4>>& T, const Vec3& center, const Vec3& dim, const std::array<int, 3>& nq)
{
    const Vec3 bottom = center - 0.5 * dim, top = center + 0.5 * dim;
    const int nx = nq[0] + 1, ny = nq[1] + 1, nz = nq[2] + 1;
    const int n_points = nx * ny * nz;
    const int nxh = nq[0], nyh = nq[1], nzh = nq[2];
    const int n_hexas = nxh * nyh * nzh;
    const double dx = (top[0] - bottom[0]) / (double)nq[0], dy = (top[1] - bottom[1]) / (double)nq[1], dz = (top[2] - bottom[2]) / (double)nq[2];
    const double hx = 0.5 * dx, hy = 0.5 * dy, hz = 0.5 * dz;
    V.resize((size_t)n_points + n_hexas);
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++)
            for (int k = 0; k < nz; k++) V[(size_t)nz * ny * i + nz * j + k] = {bottom[0] + i * dx, bottom[1] + j * dy, bottom[2] + k * dz};
    const int co = n_points;
    for (int i = 0; i < nxh; i++)
        for (int j = 0; j < nyh; j++)
            for (int k = 0; k < nzh; k++) V[(size_t)co + nzh * nyh * i + nzh * j + k] = {bottom[0] + i * dx + hx, bottom[1] + j * dy + hy, bottom[2] + k * dz + hz};
    T.clear();
    T.reserve((size_t)12 * n_hexas);
    for (int ei = 0; ei < nxh; ei++)
        for (int ej = 0; ej < nyh; ej++)
            for (int ek = 0; ek < nzh; ek++) {
                const int n[9] = {nz * ny * (ei + 0) + nz * (ej + 0) + (ek + 0), nz * ny * (ei + 0) + nz * (ej + 0) + (ek + 1), nz * ny * (ei + 0) + nz * (ej + 1) + (ek + 0),
                                  nz * ny * (ei + 0) + nz * (ej + 1) + (ek + 1), nz * ny * (ei + 1) + nz * (ej + 0) + (ek + 0), nz * ny * (ei + 1) + nz * (ej + 0) + (ek + 1),
                                  nz * ny * (ei + 1) + nz * (ej + 1) + (ek + 0), nz * ny * (ei + 1) + nz * (ej + 1) + (ek + 1), co + nzh * nyh * ei + nzh * ej + ek};
                if (((ek % 2 == 0) && (ei % 2 == ej % 2)) || ((ek % 2 == 1) && (ei % 2 != ej % 2))) {
                    const int t[12][3] = {{0, 1, 4}, {1, 5, 4}, {0, 2, 1}, {1, 2, 3}, {0, 4, 6}, {0, 6, 2}, {3, 2, 7}, {2, 6, 7}, {4, 5, 7}, {4, 7, 6}, {1, 3, 5}, {3, 7, 5}};
                    for (auto& f : t) T.push_back({n[f[0]], n[f[1]], n[f[2]], n[8]});
                } else {
                    const int t[12][3] = {{0, 1, 5}, {0, 5, 4}, {0, 3, 1}, {0, 2, 3}, {0, 4, 2}, {2, 4, 6}, {3, 2, 6}, {3, 6, 7}, {5, 7, 6}, {5, 6, 4}, {1, 3, 7}, {1, 7, 5}};
                    for (auto& f : t) T.push_back({n[f[0]], n[f[1]], n[f[2]], n[8]});
                }
            }
}

template <std::size_t N>
inline void find_edges_from_simplices(std::vector<std::array<int, 2>>& out, const std::vector<std::array<int, N>>& simplices, int n_nodes)
{
    out.clear();
    out.reserve(N * simplices.size());
    for (const auto& s : simplices)
        for (std::size_t i = 0; i < N; i++)
            for (std::size_t j = i + 1; j < N; j++) out.push_back({std::min(s[i], s[j]), std::max(s[i], s[j])});
    // the reference sorts by a[0]*n_nodes + a[1] in 32-bit int (mesh_utils.h:163), which overflows beyond 46340 nodes;
    // the 64-bit key gives the order the reference intends (and produces below that size)
    std::sort(out.begin(), out.end(), [&](const std::array<int, 2>& a, const std::array<int, 2>& b) {
        return (int64_t)a[0] * n_nodes + a[1] < (int64_t)b[0] * n_nodes + b[1];
    });
    out.erase(std::unique(out.begin(), out.end()), out.end());
}

inline void find_internal_angles(std::vector<std::array<int, 4>>& out, const std::vector<std::array<int, 3>>& tris, int n_nodes)
{
    out.clear();
    if (tris.empty()) return;
    std::vector<std::vector<int>> nn((size_t)n_nodes);
    auto push = [](std::vector<int>& v, int x) {
        if (std::find(v.begin(), v.end(), x) == v.end()) v.push_back(x);
    };
    for (const auto& t : tris)
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 3; j++) {
                push(nn[t[i]], t[j]);
                push(nn[t[j]], t[i]);
            }
    for (auto& v : nn) std::sort(v.begin(), v.end());
    std::vector<std::array<int, 2>> edges;
    find_edges_from_simplices(edges, tris, n_nodes);
    std::vector<int> buf;
    out.reserve(edges.size());
    for (const auto& e : edges) {
        buf.clear();
        std::set_intersection(nn[e[0]].begin(), nn[e[0]].end(), nn[e[1]].begin(), nn[e[1]].end(), std::back_inserter(buf));
        if (buf.size() == 2) out.push_back({e[0], e[1], buf[0], buf[1]});
        else if (buf.size() > 2) throw std::runtime_error("triangle mesh has edges with more than two incident triangles");
    }
}

// ---- small rotations (Eigen::Matrix3d / Quaterniond replacements; quaternions are (w, x, y, z)) ------------------------------------
using Mat3 = std::array<double, 9>;  // row-major
using Quat = std::array<double, 4>;
inline Mat3 mat_identity() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
inline Vec3 operator*(const Mat3& R, const Vec3& v)
{
    return {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
}
inline Mat3 transpose(const Mat3& R) { return {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]}; }
inline Mat3 matmul(const Mat3& A, const Mat3& B)
{
    Mat3 C{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    return C;
}
inline Vec3 normalized(const Vec3& v) { return (1.0 / norm(v)) * v; }
inline Quat quat_normalized(const Quat& q)
{
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    return {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
}
inline Quat quat_mul(const Quat& a, const Quat& b)  // Hamilton product
{
    return {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
}
inline Quat quat_angle_axis(double angle_rad, const Vec3& unit_axis)  // Eigen::Quaterniond(AngleAxis)
{
    const double s = std::sin(0.5 * angle_rad);
    return {std::cos(0.5 * angle_rad), s * unit_axis[0], s * unit_axis[1], s * unit_axis[2]};
}
inline Mat3 quat_to_matrix(const Quat& q)  // Eigen::Quaterniond::toRotationMatrix
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    return {1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.0 - (txx + tyy)};
}
// q1 = normalize(q0 + dt/2 (0, w) * q0)   (stark/src/models/rigidbodies/rigidbody_transformations.cpp:30-37)
inline Quat quat_time_integration(const Quat& q0, const Vec3& w, double dt)
{
    const Quat p = quat_mul({0.0, w[0], w[1], w[2]}, q0);
    return quat_normalized({q0[0] + 0.5 * dt * p[0], q0[1] + 0.5 * dt * p[1], q0[2] + 0.5 * dt * p[2], q0[3] + 0.5 * dt * p[3]});
}
inline double deg2rad(double deg) { return 2.0 * M_PI * (deg / 360.0); }  // stark/src/utils/mesh_utils.cpp:28-31
inline double rad2deg(double rad) { return 360.0 * rad / (2.0 * M_PI); }

// ---- stark::find_surface (stark/src/utils/mesh_utils.cpp:280-327) ----------------------------------------------------------------
// Faces that belong to one tet only, wound to point away from that tet, re-indexed to a compact vertex set. The reference walks an
// unordered_map, so its triangle ORDER and compact numbering are unspecified; here faces are emitted in the order of their sorted
// vertex triple and vertices are numbered by first appearance. Which faces, and the winding rule per face, are the reference's.
inline void find_surface(std::vector<std::array<int, 3>>& out_triangles, std::vector<int>& out_tri_to_tet_node, const std::vector<Vec3>& vertices,
                         const std::vector<std::array<int, 4>>& tets)
{
    // sorted face -> tet; a face of the surface occurs exactly once (sort + run-length instead of a hash map: 4 faces per tet)
    struct Face
    {
        std::array<int, 3> f;
        int tet;
    };
    std::vector<Face> all;
    all.reserve(4 * tets.size());
    for (int ti = 0; ti < (int)tets.size(); ti++) {
        const auto& t = tets[ti];
        const std::array<std::array<int, 3>, 4> faces = {{{t[0], t[1], t[2]}, {t[0], t[1], t[3]}, {t[0], t[2], t[3]}, {t[1], t[2], t[3]}}};
        for (auto f : faces) {
            std::sort(f.begin(), f.end());
            all.push_back({f, ti});
        }
    }
    std::sort(all.begin(), all.end(), [](const Face& a, const Face& b) { return a.f < b.f; });
    std::vector<std::pair<std::array<int, 3>, int>> face_tet;
    for (size_t i = 0; i < all.size();) {
        size_t j = i + 1;
        while (j < all.size() && all[j].f == all[i].f) j++;
        if ((j - i) % 2 == 1) face_tet.push_back({all[i].f, all[i].tet});  // (insert / erase toggling of the reference)
        i = j;
    }
    out_triangles.clear();
    out_tri_to_tet_node.clear();
    std::vector<int> old_to_new(vertices.size(), -1);
    for (const auto& kv : face_tet) {
        std::array<int, 3> f = kv.first;
        const auto& t = tets[kv.second];
        const Vec3 center = 0.25 * (vertices[t[0]] + vertices[t[1]] + vertices[t[2]] + vertices[t[3]]);
        // is_outward_facing (mesh_utils.cpp:20-25): normal . (tet center - face center) < 0; the reference swaps in that case (:318-320)
        const Vec3 nrm = cross(vertices[f[1]] - vertices[f[0]], vertices[f[2]] - vertices[f[0]]);
        const Vec3 fc = (1.0 / 3.0) * (vertices[f[0]] + vertices[f[1]] + vertices[f[2]]);
        if (dot(nrm, center - fc) < 0.0) std::swap(f[0], f[1]);
        std::array<int, 3> nf;
        for (int i = 0; i < 3; i++) {  // reduce_connectivity (mesh_utils.h:121-143)
            if (old_to_new[f[i]] < 0) {
                old_to_new[f[i]] = (int)out_tri_to_tet_node.size();
                out_tri_to_tet_node.push_back(f[i]);
            }
            nf[i] = old_to_new[f[i]];
        }
        out_triangles.push_back(nf);
    }
}

// ---- stark::make_box (stark/src/utils/mesh_generators.cpp:34-65): par_shapes' unit cube, centred and scaled -------------------------
inline void make_box(std::vector<Vec3>& V, std::vector<std::array<int, 3>>& T, const Vec3& size)
{
    static const double c[8][3] = {{0, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}, {1, 0, 1}};
    static const int t[12][3] = {{7, 6, 5}, {5, 4, 7}, {0, 1, 2}, {2, 3, 0}, {6, 7, 3}, {3, 2, 6}, {5, 6, 2}, {2, 1, 5}, {4, 5, 1}, {1, 0, 4}, {7, 4, 0}, {0, 3, 7}};
    V.clear();
    T.clear();
    for (auto& p : c) V.push_back({(p[0] - 0.5) * size[0], (p[1] - 0.5) * size[1], (p[2] - 0.5) * size[2]});
    for (auto& f : t) T.push_back({f[0], f[1], f[2]});
}
// stark/src/models/rigidbodies/inertia_tensors.cpp:31-47
inline Mat3 inertia_tensor_box(double mass, const Vec3& size)
{
    const double l = size[0], w = size[1], h = size[2];
    return {(1.0 / 12.0) * mass * (w * w + h * h), 0, 0, 0, (1.0 / 12.0) * mass * (l * l + h * h), 0, 0, 0, (1.0 / 12.0) * mass * (l * l + w * w)};
}

}  // namespace mistark
