// contact_geom.hpp — narrow-phase geometry of the contact detector, host/device.
//
// Restates, for one primitive pair at a time and in plain doubles:
//   * point-triangle / edge-edge closest-feature classification and squared distances
//     (TriangleMeshCollisionDetection/src/ipc_toolkit_geometry_functions.cpp:38-330, after the IPC toolkit)
//   * the edge-triangle intersection predicate (:565-585)
//   * the lagged-friction geometry: barycentric coordinates and tangent bases (stark/src/models/interactions/friction_geometry.cpp)
// The decisions (which comparison is strict, which feature wins a tie) follow the reference exactly because they decide
// which contact TABLE a pair lands in; "bit-exact contact-pair indexing" is tested against the reference's tables.
#pragma once
#include <cstdint>

#include "hdual.hpp"

namespace mistark {

struct D3
{
    double x, y, z;
};
MS_HD D3 d3(double x, double y, double z) { return D3{x, y, z}; }
MS_HD D3 operator+(const D3& a, const D3& b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MS_HD D3 operator-(const D3& a, const D3& b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MS_HD D3 operator*(double s, const D3& a) { return D3{s * a.x, s * a.y, s * a.z}; }
MS_HD double dot3(const D3& a, const D3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MS_HD D3 cross3(const D3& a, const D3& b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MS_HD double sq3(const D3& a) { return dot3(a, a); }
// Eigen's normalized(): a zero vector is returned unchanged (friction_geometry.cpp relies on it: a point exactly below its partner
// gives e x n = 0 in projection_matrix_point_point, i.e. a zero tangent basis and no friction for that contact, not a NaN)
MS_HD D3 unit3(const D3& a)
{
    const double n2 = sq3(a);
    return n2 > 0.0 ? (1.0 / ::sqrt(n2)) * a : a;
}

// closest features (ipc_toolkit_geometry_functions.h:12-40)
enum PtType : int { P_T0 = 0, P_T1, P_T2, P_E0, P_E1, P_E2, P_T };
enum EeType : int { EA0_EB0 = 0, EA0_EB1, EA1_EB0, EA1_EB1, EA_EB0, EA_EB1, EA0_EB, EA1_EB, EA_EB };

MS_HD double point_line_sq(const D3& p, const D3& e0, const D3& e1) { return sq3(cross3(e0 - p, e1 - p)) / sq3(e1 - e0); }

// coordinates of p in the basis (e1 - e0, (e1 - e0) x n) of the edge's half plane (:196-241); only the sign of the second one is used
MS_HD void edge_param(const D3& p, const D3& e0, const D3& e1, const D3& n, double& along, double& across)
{
    const D3 b0 = e1 - e0, d = p - e0;
    along = dot3(b0, d) / sq3(b0);
    across = dot3(cross3(b0, n), d);
}
// :242-273
MS_HD int point_triangle_type(const D3& p, const D3& t0, const D3& t1, const D3& t2)
{
    const D3 n = cross3(t1 - t0, t2 - t0);
    double a0, c0, a1, c1, a2, c2;
    edge_param(p, t0, t1, n, a0, c0);
    if (a0 > 0.0 && a0 < 1.0 && c0 >= 0.0) return P_E0;
    edge_param(p, t1, t2, n, a1, c1);
    if (a1 > 0.0 && a1 < 1.0 && c1 >= 0.0) return P_E1;
    edge_param(p, t2, t0, n, a2, c2);
    if (a2 > 0.0 && a2 < 1.0 && c2 >= 0.0) return P_E2;
    if (a0 <= 0.0 && a2 >= 1.0) return P_T0;
    if (a1 <= 0.0 && a0 >= 1.0) return P_T1;
    if (a2 <= 0.0 && a1 >= 1.0) return P_T2;
    return P_T;
}
// :274-303
MS_HD double point_triangle_sq_distance(int& type, const D3& p, const D3& t0, const D3& t1, const D3& t2)
{
    type = point_triangle_type(p, t0, t1, t2);
    switch (type) {
        case P_T0: return sq3(t0 - p);
        case P_T1: return sq3(t1 - p);
        case P_T2: return sq3(t2 - p);
        case P_E0: return point_line_sq(p, t0, t1);
        case P_E1: return point_line_sq(p, t1, t2);
        case P_E2: return point_line_sq(p, t2, t0);
        default: {
            const D3 n = cross3(t1 - t0, t2 - t0);
            const double h = dot3(p - t0, n);
            return h * h / sq3(n);
        }
    }
}
// :79-168 for non-parallel edges. Pairs with |u x v|^2 <= cutoff never reach a table (ProximityDetection.cpp:152-155), the
// caller drops them before asking for a type.
MS_HD int edge_edge_type(const D3& ea0, const D3& ea1, const D3& eb0, const D3& eb1)
{
    const D3 u = ea1 - ea0, v = eb1 - eb0, w = ea0 - eb0;
    const double a = sq3(u), b = dot3(u, v), c = sq3(v), d = dot3(u, w), e = dot3(v, w);
    const double D = a * c - b * b;
    const double sN = b * e - c * d;
    double tN, tD;
    int def = EA_EB;
    if (sN <= 0.0) {
        tN = e;
        tD = c;
        def = EA0_EB;
    } else if (sN >= D) {
        tN = e + b;
        tD = c;
        def = EA1_EB;
    } else {
        tN = a * e - b * d;
        tD = D;
    }
    if (tN <= 0.0) {
        if (-d <= 0.0) return EA0_EB0;
        if (-d >= a) return EA1_EB0;
        return EA_EB0;
    }
    if (tN >= tD) {
        if ((-d + b) <= 0.0) return EA0_EB1;
        if ((-d + b) >= a) return EA1_EB1;
        return EA_EB1;
    }
    return def;
}
MS_HD double edge_edge_sq_distance(int& type, const D3& ea0, const D3& ea1, const D3& eb0, const D3& eb1)
{
    type = edge_edge_type(ea0, ea1, eb0, eb1);
    switch (type) {
        case EA0_EB0: return sq3(eb0 - ea0);
        case EA0_EB1: return sq3(eb1 - ea0);
        case EA1_EB0: return sq3(eb0 - ea1);
        case EA1_EB1: return sq3(eb1 - ea1);
        case EA_EB0: return point_line_sq(eb0, ea0, ea1);
        case EA_EB1: return point_line_sq(eb1, ea0, ea1);
        case EA0_EB: return point_line_sq(ea0, eb0, eb1);
        case EA1_EB: return point_line_sq(ea1, eb0, eb1);
        default: {
            const D3 n = cross3(ea1 - ea0, eb1 - eb0);
            const double h = dot3(eb0 - ea0, n);
            return h * h / sq3(n);
        }
    }
}
// :565-585 (coplanar configurations are reported as not intersecting, as in the reference)
MS_HD bool edge_intersects_triangle(const D3& q1, const D3& q2, const D3& a, const D3& b, const D3& c)
{
    const D3 e1 = b - a, e2 = c - a;
    const D3 n = cross3(e1, e2);
    const D3 dir = q2 - q1;
    const double det = -dot3(dir, n);
    const double inv_det = 1.0 / det;
    const D3 ao = q1 - a;
    const D3 dao = cross3(ao, dir);
    const double u = dot3(e2, dao) * inv_det;
    const double v = -dot3(e1, dao) * inv_det;
    const double t = dot3(ao, n) * inv_det;
    return ::fabs(det) >= 1e-14 && t >= 0.0 && t <= 1.0 && u >= 0.0 && v >= 0.0 && (u + v) <= 1.0;
}

// ---- friction geometry (friction_geometry.cpp:4-46: barycentric coordinates after Ericson) ----------------------------------
MS_HD void bary_point_triangle(const D3& p, const D3& a, const D3& b, const D3& c, double* out)
{
    const D3 v0 = b - a, v1 = c - a, v2 = p - a;
    const double d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    const double inv_den = 1.0 / (d00 * d11 - d01 * d01);
    const double v = (d11 * d20 - d01 * d21) * inv_den;
    const double w = (d00 * d21 - d01 * d20) * inv_den;
    out[0] = 1.0 - v - w;
    out[1] = v;
    out[2] = w;
}
MS_HD void bary_point_edge(const D3& p, const D3& a, const D3& b, double* out)
{
    const D3 ab = b - a;
    const double alpha = dot3(p - a, ab) / sq3(ab);
    out[0] = 1.0 - alpha;
    out[1] = alpha;
}
MS_HD void bary_edge_edge(const D3& A, const D3& B, const D3& P, const D3& Q, double* out)
{
    const D3 da = B - A, db = Q - P, r = A - P;
    const double a = dot3(da, da), e = dot3(db, db), f = dot3(db, r), b = dot3(da, db), c = dot3(da, r);
    const double den = a * e - b * b;
    if (den < 1e-16) {  // parallel edges: the centers
        out[0] = 0.5;
        out[1] = 0.5;
        return;
    }
    const double s = (b * f - c * e) / den;
    out[0] = s;
    out[1] = (b * s + f) / e;
}
// tangent bases, rows (u, v) of the 2x3 matrix T (friction_geometry.cpp:48-101)
MS_HD void store_basis(const D3& u, const D3& v, double* T)
{
    T[0] = u.x; T[1] = u.y; T[2] = u.z;
    T[3] = v.x; T[4] = v.y; T[5] = v.z;
}
MS_HD void basis_triangle(const D3& a, const D3& b, const D3& c, double* T)
{
    const D3 v01 = a - c, v02 = b - c;
    const D3 u = unit3(v01);
    store_basis(u, unit3(cross3(cross3(v01, v02), u)), T);
}
MS_HD void basis_edge_edge(const D3& a, const D3& b, const D3& p, const D3& q, double* T)
{
    const D3 u = unit3(b - a);
    store_basis(u, unit3(cross3(u, cross3(u, q - p))), T);
}
MS_HD void basis_point_point(const D3& p, const D3& a, double* T)
{
    const D3 n = unit3(p - a);
    const D3 e = n.z < 0.99 ? d3(0.0, 0.0, 1.0) : d3(1.0, 0.0, 0.0);
    const D3 u = unit3(cross3(e, n));
    store_basis(u, unit3(cross3(u, n)), T);
}
MS_HD void basis_point_edge(const D3& p, const D3& a, const D3& b, double* T)
{
    const D3 u = unit3(b - a);
    store_basis(u, unit3(cross3(u, p - a)), T);
}

}  // namespace mistark
