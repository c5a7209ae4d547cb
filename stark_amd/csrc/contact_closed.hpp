// contact_closed.hpp — the 21 barrier and 14 friction potentials of contact_energies.hpp evaluated with hand-derived derivatives, one lane per
// contact, instead of one hyper-dual evaluation of the whole expression per pair of local DoFs (up to 300 lanes per contact).
// Definitions: stark/src/models/interactions/EnergyFrictionalContact.cpp:833-1218 (potentials), :1225-1278 (barrier, mollifier, friction),
// stark/src/models/distances.cpp:57-109 (distances), stark/src/models/rigidbodies/rigidbody_transformations.cpp (quaternion update).
//
// Every one of these potentials has the same three layers, and the chain rule is applied layer by layer:
//   1. K <= 6 POINTS P_k: positions x1 (barriers) or velocities (friction) of collision vertices. A deformable vertex is linear in its own DoF
//      block (dP/dv1 = c I with c = dt for positions, 1 for velocities). A rigid-body vertex is t0 + dt v1 + R(w1) x_loc (or v1 + w1 x R(w1) x_loc):
//      linear in the body's v1 block (c I) and nonlinear in its w1 block. Its Jacobian A_k = dP_k/dw1 and, once the point gradients g_k are
//      known, the curvature term S = d2/dw1^2 sum_k g_k . P_k(w1) come from a 3-variable second-order jet through the reference's own
//      quaternion update (rb_R1): 3 independent variables instead of all of the potential's.
//   2. M <= 3 DIFFERENCE VECTORS U_m = sum_k W_mk P_k with W in {-1, 0, 1} (friction: the barycentric weights).
//   3. The scalar: barrier(distance) x mollifier as a function psi of <= 4 INVARIANTS of the U_m — dot products, the triple product
//      U0 . (U1 x U2) and |Ua x Ub|^2 — whose gradients and Hessians with respect to the U_m are one-liners; psi's own first and second
//      derivatives come from a jet over the invariants (the reference's formulas for the distances are kept operation by operation, so the
//      point-line and point-point energies have the generic path's bits; the plane distance is (u . n)^2 / |n|^2 instead of (u . n / |n|)^2). Friction: the closed form of the C0 model in the 2-d tangent space.
// Gradient and Hessian in the local DoF blocks follow by J^T g and J^T H J + S, written block by block in the layout of the generic kernel
// (k_eval_pgh), which stays as the cross-check (option force_generic; tests compare the two at 1e-11 of the largest entry).
#pragma once
#include "contact_energies.hpp"

namespace mistark {

// ---- second-order jet in N independent variables: value, gradient, upper triangle of the Hessian -------------------------------------------
template <int N>
struct Jet
{
    static constexpr int NH = N * (N + 1) / 2;
    double v, g[N], h[NH];
    MS_HD static constexpr int hi(int i, int j) { return i * N - i * (i - 1) / 2 + (j - i); }  // i <= j
    MS_HD Jet() {}
    MS_HD Jet(double c) : v(c)
    {
#pragma unroll
        for (int i = 0; i < N; i++) g[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NH; i++) h[i] = 0.0;
    }
    MS_HD static Jet var(double c, int k)
    {
        Jet r(c);
#pragma unroll
        for (int i = 0; i < N; i++) r.g[i] = i == k ? 1.0 : 0.0;
        return r;
    }
    MS_HD double H(int i, int j) const { return i <= j ? h[hi(i, j)] : h[hi(j, i)]; }
};
template <int N>
struct promote<Jet<N>, Jet<N>> { using type = Jet<N>; };
template <int N>
struct promote<Jet<N>, double> { using type = Jet<N>; };
template <int N>
struct promote<double, Jet<N>> { using type = Jet<N>; };

template <int N>
MS_HD Jet<N> operator+(const Jet<N>& x, const Jet<N>& y)
{
    Jet<N> r;
    r.v = x.v + y.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.g[i] = x.g[i] + y.g[i];
#pragma unroll
    for (int i = 0; i < Jet<N>::NH; i++) r.h[i] = x.h[i] + y.h[i];
    return r;
}
template <int N>
MS_HD Jet<N> operator-(const Jet<N>& x, const Jet<N>& y)
{
    Jet<N> r;
    r.v = x.v - y.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.g[i] = x.g[i] - y.g[i];
#pragma unroll
    for (int i = 0; i < Jet<N>::NH; i++) r.h[i] = x.h[i] - y.h[i];
    return r;
}
template <int N>
MS_HD Jet<N> operator*(const Jet<N>& x, double s)
{
    Jet<N> r;
    r.v = x.v * s;
#pragma unroll
    for (int i = 0; i < N; i++) r.g[i] = x.g[i] * s;
#pragma unroll
    for (int i = 0; i < Jet<N>::NH; i++) r.h[i] = x.h[i] * s;
    return r;
}
template <int N>
MS_HD Jet<N> operator*(double s, const Jet<N>& x) { return x * s; }
template <int N>
MS_HD Jet<N> operator-(const Jet<N>& x) { return x * -1.0; }
template <int N>
MS_HD Jet<N> operator+(const Jet<N>& x, double s)
{
    Jet<N> r = x;
    r.v += s;
    return r;
}
template <int N>
MS_HD Jet<N> operator+(double s, const Jet<N>& x) { return x + s; }
template <int N>
MS_HD Jet<N> operator-(const Jet<N>& x, double s) { return x + (-s); }
template <int N>
MS_HD Jet<N> operator-(double s, const Jet<N>& x) { return (-x) + s; }
template <int N>
MS_HD Jet<N> operator*(const Jet<N>& x, const Jet<N>& y)
{
    Jet<N> r;
    r.v = x.v * y.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.g[i] = x.g[i] * y.v + x.v * y.g[i];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i; j < N; j++) r.h[Jet<N>::hi(i, j)] = x.h[Jet<N>::hi(i, j)] * y.v + x.v * y.h[Jet<N>::hi(i, j)] + x.g[i] * y.g[j] + x.g[j] * y.g[i];
    return r;
}
// y = f(x) given f, f', f'' at x.v
template <int N>
MS_HD Jet<N> chain(const Jet<N>& x, double f, double df, double ddf)
{
    Jet<N> r;
    r.v = f;
#pragma unroll
    for (int i = 0; i < N; i++) r.g[i] = df * x.g[i];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i; j < N; j++) r.h[Jet<N>::hi(i, j)] = df * x.h[Jet<N>::hi(i, j)] + ddf * x.g[i] * x.g[j];
    return r;
}
template <int N>
MS_HD Jet<N> inv(const Jet<N>& x)
{
    const double r = 1.0 / x.v;
    return chain(x, r, -r * r, 2.0 * r * r * r);
}
template <int N>
MS_HD Jet<N> sqrt(const Jet<N>& x)
{
    const double s = ::sqrt(x.v);
    return chain(x, s, 0.5 / s, -0.25 / (s * x.v));
}
template <int N>
MS_HD Jet<N> pow3(const Jet<N>& x) { return x * x * x; }
template <int N>
MS_HD double val(const Jet<N>& x) { return x.v; }
template <int N, class B>
MS_HD V3<Jet<N>> operator*(const Jet<N>& s, const V3<B>& b) { return {s * b.x, s * b.y, s * b.z}; }

using V3d = V3<double>;
using M3d = M3<double>;
MS_HD M3d m3_zero()
{
    M3d r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = 0.0;
    return r;
}
// r += s a b^T
MS_HD void add_outer(M3d& r, double s, const V3d& a, const V3d& b)
{
    const double ax = s * a.x, ay = s * a.y, az = s * a.z;
    r.m[0][0] += ax * b.x; r.m[0][1] += ax * b.y; r.m[0][2] += ax * b.z;
    r.m[1][0] += ay * b.x; r.m[1][1] += ay * b.y; r.m[1][2] += ay * b.z;
    r.m[2][0] += az * b.x; r.m[2][1] += az * b.y; r.m[2][2] += az * b.z;
}
MS_HD void add_diag(M3d& r, double s)
{
    r.m[0][0] += s; r.m[1][1] += s; r.m[2][2] += s;
}
// r += s [v]x   ([v]x w = v x w)
MS_HD void add_skew(M3d& r, double s, const V3d& v)
{
    r.m[0][1] -= s * v.z; r.m[0][2] += s * v.y;
    r.m[1][0] += s * v.z; r.m[1][2] -= s * v.x;
    r.m[2][0] -= s * v.y; r.m[2][1] += s * v.x;
}
MS_HD void add_scaled(M3d& r, double s, const M3d& a, bool transposed)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] += s * (transposed ? a.m[j][i] : a.m[i][j]);
}

// ---- layer 3: a scalar of <= 3 difference vectors -------------------------------------------------------------------------------------------
// gradient blocks g[m] and Hessian blocks H[(m, m')] for m <= m' (index rh(m, m')); block (m', m) is the transpose
struct Reduced
{
    V3d g[3];
    M3d H[6];
    MS_HD static constexpr int rh(int m, int n) { return m * 3 - m * (m - 1) / 2 + (n - m); }
};
// An invariant's first derivatives with respect to U_0..U_2 (zero blocks where it does not depend on a vector)
struct InvGrad
{
    V3d d[3];
};
MS_HD V3d v3_zero() { return V3d(0.0, 0.0, 0.0); }
// I = Ua . Ub  (a != b)
MS_HD double inv_dot(const V3d* U, int a, int b, InvGrad& G)
{
    G.d[0] = G.d[1] = G.d[2] = v3_zero();
    G.d[a] = U[b];
    G.d[b] = U[a];
    return dot(U[a], U[b]);
}
MS_HD void inv_dot_hess(Reduced& R, double s, int a, int b)  // a < b
{
    add_diag(R.H[Reduced::rh(a, b)], s);
}
// I = Ua . Ua
MS_HD double inv_sq(const V3d* U, int a, InvGrad& G)
{
    G.d[0] = G.d[1] = G.d[2] = v3_zero();
    G.d[a] = 2.0 * U[a];
    return dot(U[a], U[a]);
}
MS_HD void inv_sq_hess(Reduced& R, double s, int a) { add_diag(R.H[Reduced::rh(a, a)], 2.0 * s); }
// I = |Ua x Ub|^2 (a < b), evaluated as the reference does: squared norm of the cross product
MS_HD double inv_cross2(const V3d* U, int a, int b, InvGrad& G)
{
    const V3d c = cross(U[a], U[b]);
    G.d[0] = G.d[1] = G.d[2] = v3_zero();
    G.d[a] = 2.0 * cross(U[b], c);
    G.d[b] = 2.0 * cross(c, U[a]);
    return dot(c, c);
}
MS_HD void inv_cross2_hess(Reduced& R, double s, const V3d* U, int a, int b)  // a < b
{
    const V3d& p = U[a];
    const V3d& q = U[b];
    M3d& Haa = R.H[Reduced::rh(a, a)];
    add_diag(Haa, 2.0 * s * dot(q, q));
    add_outer(Haa, -2.0 * s, q, q);
    M3d& Hbb = R.H[Reduced::rh(b, b)];
    add_diag(Hbb, 2.0 * s * dot(p, p));
    add_outer(Hbb, -2.0 * s, p, p);
    M3d& Hab = R.H[Reduced::rh(a, b)];  // d(grad_a)/d(Ub) = 2 (2 p q^T - q p^T - (p.q) I)
    add_outer(Hab, 4.0 * s, p, q);
    add_outer(Hab, -2.0 * s, q, p);
    add_diag(Hab, -2.0 * s * dot(p, q));
}
// I = U0 . (U1 x U2)
MS_HD double inv_triple(const V3d* U, InvGrad& G)
{
    const V3d n = cross(U[1], U[2]);
    G.d[0] = n;
    G.d[1] = cross(U[2], U[0]);
    G.d[2] = cross(U[0], U[1]);
    return dot(U[0], n);
}
MS_HD void inv_triple_hess(Reduced& R, double s, const V3d* U)
{
    add_skew(R.H[Reduced::rh(0, 1)], -s, U[2]);  // d(U1 x U2)/dU1 = -[U2]x
    add_skew(R.H[Reduced::rh(0, 2)], s, U[1]);   // d(U1 x U2)/dU2 = [U1]x
    add_skew(R.H[Reduced::rh(1, 2)], -s, U[0]);  // d(U2 x U0)/dU2 = -[U0]x
}
// psi(I_0..I_{NI-1}) -> gradient and the rank terms of the Hessian (the callers add psi_i d2I_i)
template <int NI>
MS_HD void reduce_first_and_rank(const Jet<NI>& psi, const InvGrad* G, Reduced& R)
{
#pragma unroll
    for (int m = 0; m < 3; m++) {
        V3d a = v3_zero();
#pragma unroll
        for (int i = 0; i < NI; i++) a = a + psi.g[i] * G[i].d[m];
        R.g[m] = a;
    }
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = m; n < 3; n++) {
            M3d B = m3_zero();
#pragma unroll
            for (int i = 0; i < NI; i++) {
                V3d t = v3_zero();  // sum_j psi_ij dI_j[n]
#pragma unroll
                for (int j = 0; j < NI; j++) t = t + psi.H(i, j) * G[j].d[n];
                add_outer(B, 1.0, G[i].d[m], t);
            }
            R.H[Reduced::rh(m, n)] = B;
        }
}
template <class T>
MS_HD T barrier_of_sq(const T& d2, double dhat, double k) { return barrier(sqrt(d2), dhat, k); }
template <class T>
MS_HD T mollifier_of(const T& x, double eps_x)
{
    if (val(x) > eps_x) return T(1.0);
    const T r = x * (1.0 / eps_x);
    return (2.0 - r) * r;
}
// DIST: 0 point-point |U0|, 1 point-line (U0 from the line's first point, U1 the line), 2 plane / line-line U0 . (U1 x U2) / |U1 x U2|.
// MOLL: multiplied by the edge-edge mollifier of |U1 x U2|^2 (DIST 0, 2) or |U1 x U2|^2 with U2 the other edge (DIST 1).
template <int DIST, bool MOLL>
MS_HD double contact_scalar(const V3d* U, double dhat, double k, double eps_x, Reduced& R)
{
    if constexpr (DIST == 0 && !MOLL) {
        InvGrad G[1];
        const Jet<1> I0 = Jet<1>::var(inv_sq(U, 0, G[0]), 0);
        const Jet<1> psi = barrier_of_sq(I0, dhat, k);
        reduce_first_and_rank<1>(psi, G, R);
        inv_sq_hess(R, psi.g[0], 0);
        return psi.v;
    } else if constexpr (DIST == 0 && MOLL) {
        InvGrad G[2];
        const Jet<2> I0 = Jet<2>::var(inv_sq(U, 0, G[0]), 0);
        const Jet<2> I1 = Jet<2>::var(inv_cross2(U, 1, 2, G[1]), 1);
        const Jet<2> psi = mollifier_of(I1, eps_x) * barrier_of_sq(I0, dhat, k);
        reduce_first_and_rank<2>(psi, G, R);
        inv_sq_hess(R, psi.g[0], 0);
        inv_cross2_hess(R, psi.g[1], U, 1, 2);
        return psi.v;
    } else if constexpr (DIST == 1) {
        constexpr int NI = MOLL ? 4 : 3;
        InvGrad G[NI];
        const Jet<NI> uu = Jet<NI>::var(inv_sq(U, 0, G[0]), 0);
        const Jet<NI> ue = Jet<NI>::var(inv_dot(U, 0, 1, G[1]), 1);
        const Jet<NI> ee = Jet<NI>::var(inv_sq(U, 1, G[2]), 2);
        Jet<NI> psi = barrier_of_sq(uu - ue * ue * inv(ee), dhat, k);  // sq_distance_point_line (distances.cpp:61-68)
        if constexpr (MOLL) {
            const Jet<NI> x = Jet<NI>::var(inv_cross2(U, 1, 2, G[3]), 3);
            psi = mollifier_of(x, eps_x) * psi;
        }
        reduce_first_and_rank<NI>(psi, G, R);
        inv_sq_hess(R, psi.g[0], 0);
        inv_dot_hess(R, psi.g[1], 0, 1);
        inv_sq_hess(R, psi.g[2], 1);
        if constexpr (MOLL) inv_cross2_hess(R, psi.g[3], U, 1, 2);
        return psi.v;
    } else {
        InvGrad G[2];
        const Jet<2> l = Jet<2>::var(inv_triple(U, G[0]), 0);
        const Jet<2> x = Jet<2>::var(inv_cross2(U, 1, 2, G[1]), 1);
        Jet<2> psi = barrier_of_sq(l * l * inv(x), dhat, k);
        if constexpr (MOLL) psi = mollifier_of(x, eps_x) * psi;
        reduce_first_and_rank<2>(psi, G, R);
        inv_triple_hess(R, psi.g[0], U);
        inv_cross2_hess(R, psi.g[1], U, 1, 2);
        return psi.v;
    }
}

// ---- layer 1: points ---------------------------------------------------------------------------------------------------------------------------
// K vertices of a deformable object at in[o]: positions (v1[K], x0[K], dt: 6K + 1 inputs) or velocities (v1[K]: 3K inputs)
template <int K, bool VEL>
MS_HD void soft_points(const double* in, int o, V3d* P)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        const V3d v1(in[o + 3 * k], in[o + 3 * k + 1], in[o + 3 * k + 2]);
        if (VEL) P[k] = v1;
        else {
            const V3d x0(in[o + 3 * K + 3 * k], in[o + 3 * K + 3 * k + 1], in[o + 3 * K + 3 * k + 2]);
            P[k] = x0 + in[o + 6 * K] * v1;
        }
    }
}
template <int K, bool VEL>
constexpr int soft_len() { return VEL ? 3 * K : 6 * K + 1; }
// K vertices of one rigid body at in[o]: dt, x_loc[K], v1, w1, t0, q0 (3K + 14 inputs)
template <int K>
constexpr int rb_len() { return 3 * K + 14; }
template <int K, bool VEL>
MS_HD void rb_jets(const double* in, int o, V3<Jet<3>>* p)
{
    const double dt = in[o];
    const int ob = o + 1 + 3 * K;
    const V3<Jet<3>> w(Jet<3>::var(in[ob + 3], 0), Jet<3>::var(in[ob + 4], 1), Jet<3>::var(in[ob + 5], 2));
    const M3<Jet<3>> R1 = rb_R1(in + ob + 9, w, dt);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const V3d xl(in[o + 1 + 3 * k], in[o + 2 + 3 * k], in[o + 3 + 3 * k]);
        const V3<Jet<3>> r = R1 * xl;
        if (VEL) p[k] = cross(w, r);
        else p[k] = r;
    }
}
// values and dP/dw1 (the v1 block's Jacobian is c I)
template <int K, bool VEL>
MS_HD void rb_points(const double* in, int o, V3d* P, M3d* A)
{
    V3<Jet<3>> p[K];
    rb_jets<K, VEL>(in, o, p);
    const double dt = in[o];
    const int ob = o + 1 + 3 * K;
    const V3d v1(in[ob], in[ob + 1], in[ob + 2]), t0(in[ob + 6], in[ob + 7], in[ob + 8]);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const V3d r(p[k].x.v, p[k].y.v, p[k].z.v);
        if (VEL) P[k] = v1 + r;
        else P[k] = (t0 + dt * v1) + r;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[k].m[0][j] = p[k].x.g[j];
            A[k].m[1][j] = p[k].y.g[j];
            A[k].m[2][j] = p[k].z.g[j];
        }
    }
}
// S = d2/dw1^2 sum_k g_k . P_k(w1)
template <int K, bool VEL>
MS_HD M3d rb_curvature(const double* in, int o, const V3d* g)
{
    V3<Jet<3>> p[K];
    rb_jets<K, VEL>(in, o, p);
    Jet<3> s = dot(g[0], p[0]);
#pragma unroll
    for (int k = 1; k < K; k++) s = s + dot(g[k], p[k]);
    M3d S;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) S.m[i][j] = s.H(i, j);
    return S;
}

// ---- the potentials' structure, at compile time ------------------------------------------------------------------------------------------------
// One GROUP = the vertices bound by one getter call: K points, deformable (K own blocks) or one rigid body (a v1 block and a w1 block).
// Local block order (SecondOrderCompiledPotential.cpp:10-33): deformable blocks in reading order, rigid v1 blocks, rigid w1 blocks.
struct GroupDesc
{
    int rigid, K, first_point, first_block /*deformable: block of point 0; rigid: slot*/, in_off;
};
template <int NG>
struct Structure
{
    GroupDesc g[NG];
    int NS, NR, NPT, in_end;
};
// groups given as (rigid, K, extra inputs that follow the group: rest positions of an edge) triples
template <int NG, bool VEL>
constexpr Structure<NG> make_structure(const int (&rigid)[NG], const int (&K)[NG], const int (&skip)[NG])
{
    Structure<NG> s{};
    int o = 0, is = 0, ir = 0, np = 0;
    for (int i = 0; i < NG; i++) {
        s.g[i].rigid = rigid[i];
        s.g[i].K = K[i];
        s.g[i].first_point = np;
        s.g[i].in_off = o;
        if (rigid[i]) {
            s.g[i].first_block = ir++;
            o += 3 * K[i] + 14;
        } else {
            s.g[i].first_block = is;
            is += K[i];
            o += VEL ? 3 * K[i] : 6 * K[i] + 1;
        }
        o += skip[i];
        np += K[i];
    }
    s.NS = is;
    s.NR = ir;
    s.NPT = np;
    s.in_end = o;
    return s;
}

// `Out` (kernels.hip: ClosedOut) receives the element's numbers: put_grad(local block, V3d) and put_block(NB, ba, bb, M3d) — the latter
// writes block (ba, bb) and its transpose (bb, ba).

// ---- layers 2 -> 1 -> DoF blocks ------------------------------------------------------------------------------------------------------------------
// ST: a struct with static constexpr members `S` (Structure<NG>), NG, M, VEL and a static W(m, k) (compile-time signs for the barriers,
// run-time weights for friction come through `wr`).
// Point-level Hessian block (k, k') = sum_mm' W_mk W_m'k' Hr(m, m')
template <class ST>
MS_HD M3d point_block(const Reduced& R, const double* wr, int k, int kp)
{
    M3d B = m3_zero();
#pragma unroll
    for (int m = 0; m < ST::M; m++)
#pragma unroll
        for (int n = 0; n < ST::M; n++) {
            const double w = ST::weight(wr, m, k) * ST::weight(wr, n, kp);
            if (ST::weight_is_zero(m, k) || ST::weight_is_zero(n, kp)) continue;
            if (m <= n) add_scaled(B, w, R.H[Reduced::rh(m, n)], false);
            else add_scaled(B, w, R.H[Reduced::rh(n, m)], true);
        }
    return B;
}
MS_HD M3d mul_AtB(const M3d& A, const M3d& B)  // A^T B
{
    M3d r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = A.m[0][i] * B.m[0][j] + A.m[1][i] * B.m[1][j] + A.m[2][i] * B.m[2][j];
    return r;
}
MS_HD void add_m3(M3d& r, const M3d& a)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] += a.m[i][j];
}
MS_HD M3d scaled_m3(double s, const M3d& a)
{
    M3d r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
    return r;
}
MS_HD V3d mul_Atv(const M3d& A, const V3d& v)
{
    return V3d(A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
               A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z);
}

// Everything after the scalar: point gradients, curvature of the rigid kinematics, gradient and Hessian blocks in local DoF order.
// P: points, A: dP/dw1 of rigid points, R: gradient / Hessian with respect to the difference vectors, wr: run-time weights (friction)
template <class ST, bool STORE_H, class Out>
MS_HD void scatter_to_dofs(const double* in, const M3d* A, const Reduced& R, const double* wr, double c, const Out& out)
{
    constexpr auto S = ST::structure();
    constexpr int NG = ST::NG, NB = S.NS + 2 * S.NR;
    V3d gp[S.NPT];  // point gradients
#pragma unroll
    for (int k = 0; k < S.NPT; k++) {
        V3d a = v3_zero();
#pragma unroll
        for (int m = 0; m < ST::M; m++)
            if (!ST::weight_is_zero(m, k)) a = a + ST::weight(wr, m, k) * R.g[m];
        gp[k] = a;
    }
    // gradient
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
        const GroupDesc G = S.g[gi];
        if (!G.rigid) {
#pragma unroll
            for (int k = 0; k < G.K; k++) out.put_grad(G.first_block + k, c * gp[G.first_point + k]);
        } else {
            V3d gv = v3_zero(), gw = v3_zero();
#pragma unroll
            for (int k = 0; k < G.K; k++) {
                gv = gv + gp[G.first_point + k];
                gw = gw + mul_Atv(A[G.first_point + k], gp[G.first_point + k]);
            }
            out.put_grad(S.NS + G.first_block, c * gv);
            out.put_grad(S.NS + S.NR + G.first_block, gw);
        }
    }
    if (!STORE_H) return;
    // Hessian, group pair by group pair (gi <= gj)
#pragma unroll
    for (int gi = 0; gi < NG; gi++)
#pragma unroll
        for (int gj = gi; gj < NG; gj++) {
            const GroupDesc Ga = S.g[gi], Gb = S.g[gj];
            if (!Ga.rigid && !Gb.rigid) {
#pragma unroll
                for (int k = 0; k < Ga.K; k++)
#pragma unroll
                    for (int l = (gi == gj ? k : 0); l < Gb.K; l++)
                        out.put_block(NB, Ga.first_block + k, Gb.first_block + l, scaled_m3(c * c, point_block<ST>(R, wr, Ga.first_point + k, Gb.first_point + l)));
            } else if (!Ga.rigid && Gb.rigid) {
#pragma unroll
                for (int k = 0; k < Ga.K; k++) {
                    M3d Hv = m3_zero(), Hw = m3_zero();
#pragma unroll
                    for (int l = 0; l < Gb.K; l++) {
                        const M3d B = point_block<ST>(R, wr, Ga.first_point + k, Gb.first_point + l);
                        add_m3(Hv, B);
                        add_m3(Hw, B * A[Gb.first_point + l]);
                    }
                    out.put_block(NB, Ga.first_block + k, S.NS + Gb.first_block, scaled_m3(c * c, Hv));
                    out.put_block(NB, Ga.first_block + k, S.NS + S.NR + Gb.first_block, scaled_m3(c, Hw));
                }
            } else if (Ga.rigid && !Gb.rigid) {  // (a deformable group behind a rigid one: blocks (soft, rigid) are written from the transposes)
#pragma unroll
                for (int l = 0; l < Gb.K; l++) {
                    M3d Hv = m3_zero(), Hw = m3_zero();
#pragma unroll
                    for (int k = 0; k < Ga.K; k++) {
                        const M3d B = point_block<ST>(R, wr, Gb.first_point + l, Ga.first_point + k);
                        add_m3(Hv, B);
                        add_m3(Hw, B * A[Ga.first_point + k]);
                    }
                    out.put_block(NB, Gb.first_block + l, S.NS + Ga.first_block, scaled_m3(c * c, Hv));
                    out.put_block(NB, Gb.first_block + l, S.NS + S.NR + Ga.first_block, scaled_m3(c, Hw));
                }
            } else {
                M3d Hvv = m3_zero(), Hvw = m3_zero(), Hwv = m3_zero(), Hww = m3_zero();
#pragma unroll
                for (int k = 0; k < Ga.K; k++)
#pragma unroll
                    for (int l = 0; l < Gb.K; l++) {
                        const M3d B = point_block<ST>(R, wr, Ga.first_point + k, Gb.first_point + l);
                        const M3d BA = B * A[Gb.first_point + l];
                        add_m3(Hvv, B);
                        add_m3(Hvw, BA);
                        add_m3(Hwv, mul_AtB(A[Ga.first_point + k], B));
                        add_m3(Hww, mul_AtB(A[Ga.first_point + k], BA));
                    }
                const int va = S.NS + Ga.first_block, wa = S.NS + S.NR + Ga.first_block, vb = S.NS + Gb.first_block, wb = S.NS + S.NR + Gb.first_block;
                if (gi == gj) {
                    M3d Sg;
                    if (Ga.K == 1) Sg = rb_curvature<1, ST::VEL>(in, Ga.in_off, gp + Ga.first_point);
                    else if (Ga.K == 2) Sg = rb_curvature<2, ST::VEL>(in, Ga.in_off, gp + Ga.first_point);
                    else Sg = rb_curvature<3, ST::VEL>(in, Ga.in_off, gp + Ga.first_point);
                    add_m3(Hww, Sg);
                    out.put_block(NB, va, va, scaled_m3(c * c, Hvv));
                    out.put_block(NB, va, wa, scaled_m3(c, Hvw));
                    out.put_block(NB, wa, wa, Hww);
                } else {
                    out.put_block(NB, va, vb, scaled_m3(c * c, Hvv));
                    out.put_block(NB, va, wb, scaled_m3(c, Hvw));
                    out.put_block(NB, vb, wa, scaled_m3(c, transpose(Hwv)));  // (vb < wa: v blocks come before w blocks)
                    out.put_block(NB, wa, wb, Hww);
                }
            }
        }
}
// all points of a potential (values, rigid Jacobians)
template <class ST>
MS_HD void read_points(const double* in, V3d* P, M3d* A)
{
    constexpr auto S = ST::structure();
#pragma unroll
    for (int gi = 0; gi < ST::NG; gi++) {
        const GroupDesc G = S.g[gi];
        if (!G.rigid) {
            if (G.K == 1) soft_points<1, ST::VEL>(in, G.in_off, P + G.first_point);
            else if (G.K == 2) soft_points<2, ST::VEL>(in, G.in_off, P + G.first_point);
            else soft_points<3, ST::VEL>(in, G.in_off, P + G.first_point);
        } else {
            if (G.K == 1) rb_points<1, ST::VEL>(in, G.in_off, P + G.first_point, A + G.first_point);
            else if (G.K == 2) rb_points<2, ST::VEL>(in, G.in_off, P + G.first_point, A + G.first_point);
            else rb_points<3, ST::VEL>(in, G.in_off, P + G.first_point, A + G.first_point);
        }
    }
}

// ---- the three families ---------------------------------------------------------------------------------------------------------------------------
template <Src SA, int KA, Src SB, int KB, int DIST>
struct PT_Closed
{
    static constexpr int NG = 2, M = DIST == 0 ? 1 : ((DIST == 1 || DIST == 3) ? 2 : 3);
    static constexpr bool VEL = false;
    static constexpr Structure<2> structure()
    {
        const int rigid_[2] = {SA == SRB, SB == SRB}, K_[2] = {KA, KB}, skip_[2] = {0, 0};
        return make_structure<2, false>(rigid_, K_, skip_);
    }
    // points: a[0..KA) then b[0..KB)
    MS_HD static constexpr int sign(int m, int k)
    {
        const int a0 = 0, b0 = KA;
        if (DIST == 0) return k == a0 ? 1 : (k == b0 ? -1 : 0);                                   // a0 - b0
        if (DIST == 1) return m == 0 ? (k == a0 ? 1 : (k == b0 ? -1 : 0)) : (k == b0 + 1 ? 1 : (k == b0 ? -1 : 0));
        if (DIST == 2) return m == 0 ? (k == a0 ? 1 : (k == b0 ? -1 : 0)) : (m == 1 ? (k == b0 ? 1 : (k == b0 + 2 ? -1 : 0)) : (k == b0 + 1 ? 1 : (k == b0 + 2 ? -1 : 0)));
        if (DIST == 3) return m == 0 ? (k == b0 ? 1 : (k == a0 ? -1 : 0)) : (k == a0 + 1 ? 1 : (k == a0 ? -1 : 0));
        return m == 0 ? (k == b0 ? 1 : (k == a0 ? -1 : 0)) : (m == 1 ? (k == a0 ? 1 : (k == a0 + 2 ? -1 : 0)) : (k == a0 + 1 ? 1 : (k == a0 + 2 ? -1 : 0)));
    }
    MS_HD static constexpr bool weight_is_zero(int m, int k) { return sign(m, k) == 0; }
    MS_HD static double weight(const double*, int m, int k) { return (double)sign(m, k); }
    template <bool STORE_H, class Out>
    MS_HD static double eval(const double* in, const Out& out)
    {
        constexpr auto S = structure();
        V3d P[S.NPT];
        M3d A[S.NPT];
        read_points<PT_Closed>(in, P, A);
        const double dhat = in[S.in_end] + in[S.in_end + 1], k = in[S.in_end + 2];
        V3d U[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            V3d u = v3_zero();
#pragma unroll
            for (int q = 0; q < S.NPT; q++)  // (the positive point first: the same subtraction as the reference's p - a)
                if (m < M && sign(m, q) > 0) u = P[q];
#pragma unroll
            for (int q = 0; q < S.NPT; q++)
                if (m < M && sign(m, q) < 0) u = u - P[q];
            U[m] = u;
        }
        Reduced R;
        const double E = contact_scalar<(DIST == 0 ? 0 : ((DIST == 1 || DIST == 3) ? 1 : 2)), false>(U, dhat, k, 0.0, R);
        scatter_to_dofs<PT_Closed, STORE_H>(in, A, R, nullptr, in[dt_offset()], out);
        return E;
    }
    MS_HD static constexpr int dt_offset() { return structure().g[0].rigid ? structure().g[0].in_off : structure().g[0].in_off + 6 * structure().g[0].K; }
};
// groups: edge a, [point p], edge b, [point q]; an edge is followed by its two rest positions (6 inputs)
template <Src SA, bool PA_, Src SB, bool PB_, int DIST>
struct EE_Closed
{
    static constexpr int NG = 2 + (PA_ ? 1 : 0) + (PB_ ? 1 : 0), M = 3;
    static constexpr bool VEL = false;
    static constexpr int gEA = 0, gP = PA_ ? 1 : -1, gEB = PA_ ? 2 : 1, gQ = PB_ ? gEB + 1 : -1;
    struct Lists
    {
        int rigid[NG], K[NG], skip[NG];
    };
    static constexpr Lists lists()
    {
        Lists l{};
        int i = 0;
        l.rigid[i] = SA == SRB; l.K[i] = 2; l.skip[i] = 6; i++;
        if (PA_) { l.rigid[i] = SA == SRB; l.K[i] = 1; l.skip[i] = 0; i++; }
        l.rigid[i] = SB == SRB; l.K[i] = 2; l.skip[i] = 6; i++;
        if (PB_) { l.rigid[i] = SB == SRB; l.K[i] = 1; l.skip[i] = 0; i++; }
        return l;
    }
    static constexpr Structure<NG> structure()
    {
        const Lists l = lists();
        return make_structure<NG, false>(l.rigid, l.K, l.skip);
    }
    static constexpr int ea0 = 0, ea1 = 1, pp = PA_ ? 2 : -1, eb0 = PA_ ? 3 : 2, eb1 = eb0 + 1, qq = PB_ ? eb0 + 2 : -1;
    MS_HD static constexpr int pm(int k, int plus, int minus) { return k == plus ? 1 : (k == minus ? -1 : 0); }
    MS_HD static constexpr int sign(int m, int k)
    {
        if (DIST == 0) return m == 0 ? pm(k, pp, qq) : (m == 1 ? pm(k, ea1, ea0) : pm(k, eb1, eb0));   // |p - q|, mollifier (ea, eb)
        if (DIST == 1) return m == 0 ? pm(k, pp, eb0) : (m == 1 ? pm(k, eb1, eb0) : pm(k, ea1, ea0));  // p to line eb, mollifier (eb, ea)
        if (DIST == 2) return m == 0 ? pm(k, eb0, ea0) : (m == 1 ? pm(k, ea1, ea0) : pm(k, eb1, eb0)); // line-line
        return m == 0 ? pm(k, qq, ea0) : (m == 1 ? pm(k, ea1, ea0) : pm(k, eb1, eb0));                 // q to line ea, mollifier (ea, eb)
    }
    MS_HD static constexpr bool weight_is_zero(int m, int k) { return sign(m, k) == 0; }
    MS_HD static double weight(const double*, int m, int k) { return (double)sign(m, k); }
    template <bool STORE_H, class Out>
    MS_HD static double eval(const double* in, const Out& out)
    {
        constexpr auto S = structure();
        V3d P[S.NPT];
        M3d A[S.NPT];
        read_points<EE_Closed>(in, P, A);
        const double k = in[S.in_end], dhat = in[S.in_end + 1] + in[S.in_end + 2];
        // rest positions behind the two edge groups
        constexpr int ra = S.g[gEA].in_off + (S.g[gEA].rigid ? rb_len<2>() : soft_len<2, false>());
        constexpr int rb = S.g[gEB].in_off + (S.g[gEB].rigid ? rb_len<2>() : soft_len<2, false>());
        const V3d ra0(in[ra], in[ra + 1], in[ra + 2]), ra1(in[ra + 3], in[ra + 4], in[ra + 5]), rb0(in[rb], in[rb + 1], in[rb + 2]), rb1(in[rb + 3], in[rb + 4], in[rb + 5]);
        const double eps_x = 1e-3 * sqnorm(ra0 - ra1) * sqnorm(rb0 - rb1);
        V3d U[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            V3d u = v3_zero();
#pragma unroll
            for (int q = 0; q < S.NPT; q++)
                if (sign(m, q) > 0) u = P[q];
#pragma unroll
            for (int q = 0; q < S.NPT; q++)
                if (sign(m, q) < 0) u = u - P[q];
            U[m] = u;
        }
        Reduced R;
        const double E = contact_scalar<(DIST == 0 ? 0 : (DIST == 2 ? 2 : 1)), true>(U, dhat, k, eps_x, R);
        scatter_to_dofs<EE_Closed, STORE_H>(in, A, R, nullptr, in[dt_offset()], out);
        return E;
    }
    MS_HD static constexpr int dt_offset() { return structure().g[0].rigid ? structure().g[0].in_off : structure().g[0].in_off + 12; }
};
// friction: one difference vector v = sum_k w_k velocity_k with the barycentric weights of the lagged contact
template <Src SA, int KA, Src SB, int KB, int KIND>
struct Friction_Closed
{
    static constexpr int NG = 2, M = 1;
    static constexpr bool VEL = true;
    static constexpr int NBARY = KIND == 0 ? 0 : ((KIND == 2 || KIND == 5) ? 3 : 2);
    static constexpr Structure<2> structure()
    {
        const int rigid_[2] = {SA == SRB, SB == SRB}, K_[2] = {KA, KB}, skip_[2] = {0, 0};
        return make_structure<2, true>(rigid_, K_, skip_);
    }
    MS_HD static constexpr bool weight_is_zero(int, int) { return false; }
    MS_HD static double weight(const double* wr, int, int k) { return wr[k]; }
    template <bool STORE_H, class Out>
    MS_HD static double eval(const double* in, const Out& out)
    {
        constexpr auto S = structure();
        V3d P[S.NPT];
        M3d A[S.NPT];
        read_points<Friction_Closed>(in, P, A);
        int o = S.in_end;
        double bary[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NBARY; i++) bary[i] = in[o + i];
        o += NBARY;
        const V3d* a = P;
        const V3d* b = P + KA;
        double w[KA + KB];
        V3d v;  // the reference's expressions, operation by operation (contact_energies.hpp: Friction::energy)
        if (KIND == 0) { v = b[0] - a[0]; w[0] = -1.0; w[KA] = 1.0; }
        else if (KIND == 1) { v = (bary[0] * b[0] + bary[1] * b[KB > 1 ? 1 : 0]) - a[0]; w[0] = -1.0; w[KA] = bary[0]; w[KA + (KB > 1 ? 1 : 0)] = bary[1]; }
        else if (KIND == 2) {
            v = (bary[0] * b[0] + bary[1] * b[KB > 1 ? 1 : 0] + bary[2] * b[KB > 2 ? 2 : 0]) - a[0];
            w[0] = -1.0; w[KA] = bary[0]; w[KA + (KB > 1 ? 1 : 0)] = bary[1]; w[KA + (KB > 2 ? 2 : 0)] = bary[2];
        } else if (KIND == 3) {
            v = (b[0] + bary[1] * (b[KB > 1 ? 1 : 0] - b[0])) - (a[0] + bary[0] * (a[KA > 1 ? 1 : 0] - a[0]));
            w[0] = -(1.0 - bary[0]); w[KA > 1 ? 1 : 0] = -bary[0]; w[KA] = 1.0 - bary[1]; w[KA + (KB > 1 ? 1 : 0)] = bary[1];
        } else if (KIND == 4) { v = (bary[0] * a[0] + bary[1] * a[KA > 1 ? 1 : 0]) - b[0]; w[0] = bary[0]; w[KA > 1 ? 1 : 0] = bary[1]; w[KA] = -1.0; }
        else {
            v = (bary[0] * a[0] + bary[1] * a[KA > 1 ? 1 : 0] + bary[2] * a[KA > 2 ? 2 : 0]) - b[0];
            w[0] = bary[0]; w[KA > 1 ? 1 : 0] = bary[1]; w[KA > 2 ? 2 : 0] = bary[2]; w[KA] = -1.0;
        }
        // C0 friction in the tangent plane (EnergyFrictionalContact.cpp:1260-1278, :1321-1329): tail T (2x3), mu, fn, epsv, dt
        const double* Tm = in + o;
        const double mu = in[o + 6], fn = in[o + 7], epsv = in[o + 8], dt = in[o + 9];
        const double ut0 = (Tm[0] * v.x + Tm[1] * v.y + Tm[2] * v.z) * dt + 1.13e-9;
        const double ut1 = (Tm[3] * v.x + Tm[4] * v.y + Tm[5] * v.z) * dt - 1.07e-9;
        const double u = ::sqrt(ut0 * ut0 + ut1 * ut1);
        const double epsu = dt * epsv;
        const double kf = mu * fn / epsu;
        const double eps = mu * fn / (2.0 * kf);
        double E, h00, h01, h11, g0, g1;  // derivatives with respect to (ut0, ut1)
        if (u < epsu) {
            E = (0.5 * kf) * (u * u);
            g0 = kf * ut0; g1 = kf * ut1;
            h00 = kf; h01 = 0.0; h11 = kf;
        } else {
            E = (mu * fn) * (u - eps);
            const double s = mu * fn / u, n0 = ut0 / u, n1 = ut1 / u;
            g0 = s * ut0; g1 = s * ut1;
            h00 = s * (1.0 - n0 * n0); h01 = -s * n0 * n1; h11 = s * (1.0 - n1 * n1);
        }
        const V3d t0(Tm[0], Tm[1], Tm[2]), t1(Tm[3], Tm[4], Tm[5]);
        Reduced R;
        R.g[0] = dt * (g0 * t0 + g1 * t1);
        M3d H = m3_zero();
        add_outer(H, dt * dt * h00, t0, t0);
        add_outer(H, dt * dt * h01, t0, t1);
        add_outer(H, dt * dt * h01, t1, t0);
        add_outer(H, dt * dt * h11, t1, t1);
        R.H[0] = H;
        scatter_to_dofs<Friction_Closed, STORE_H>(in, A, R, w, 1.0, out);
        return E;
    }
};

// which potentials have a closed form: the family a potential struct derives from
template <Src A, int KA, Src B, int KB, int DIST>
MS_HD PT_Closed<A, KA, B, KB, DIST> closed_of(const PT_Contact<A, KA, B, KB, DIST>*);
template <Src A, bool PA_, Src B, bool PB_, int DIST>
MS_HD EE_Closed<A, PA_, B, PB_, DIST> closed_of(const EE_Contact<A, PA_, B, PB_, DIST>*);
template <Src A, int KA, Src B, int KB, int KIND>
MS_HD Friction_Closed<A, KA, B, KB, KIND> closed_of(const Friction<A, KA, B, KB, KIND>*);
MS_HD void closed_of(...);
template <class En>
using closed_t = decltype(closed_of((const En*)nullptr));
template <class En>
constexpr bool has_closed_contact = !std::is_void<closed_t<En>>::value;

}  // namespace mistark
