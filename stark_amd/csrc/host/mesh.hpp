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

}  // namespace mistark
