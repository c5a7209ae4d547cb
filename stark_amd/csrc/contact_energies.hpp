// contact_energies.hpp — IPC barrier (21) and lagged friction (14) potentials of
// stark/src/models/interactions/EnergyFrictionalContact.cpp, written once against a generic scalar (see energies.hpp).
//
// The reference builds each of these potentials from a few "symbol getter" helpers, each of which binds a fixed group of
// arrays (EnergyFrictionalContact.cpp:1358-1423). The same decomposition is used here: a potential is a sequence of PARTS
// read by a cursor in the reference's binding order, followed by a tail.
//   D_X1<K>  : _get_d_x1   v1[K]*, x0[K], dt                       -> x1 = x0 + dt v1                      (:1393-1399)
//   D_X<K>   : _get_d_X    X[K]                                    -> rest positions                       (:1400-1403)
//   RB_X1<K> : _get_rb_x1  dt, x_loc[K], v1*, w1*, t0, q0_         -> x1 = t0 + dt v1 + R(q1) x_loc        (:1364-1369, RigidBodyDynamics.cpp:46-62)
//   RB_X<K>  : _get_rb_X   x_loc[K]                                                                       (:1370-1373)
//   D_V1<K>  : _get_d_v1   v1[K]*                                                                         (:1389-1392)
//   RB_V1<K> : _get_rb_v1  dt, x_loc[K], v1*, w1*, t0, q0_         -> v1 + w1 x (R(q1) x_loc)              (:1358-1363, RigidBodyDynamics.cpp:67-85)
// A rigid body fetched twice by the reference (edge + point getters) contributes two independent pairs of DoF blocks that
// map to the same global rows, exactly as in the reference (the assembly sums them).
// Local DoF order: soft blocks in reading order, then the rigid v1 blocks, then the rigid w1 blocks
// (DoF sets in registration order, SecondOrderCompiledPotential.cpp:10-33).
#pragma once
#include "energies.hpp"

namespace mistark {

// ---- compile-time stride lists --------------------------------------------------------------------------------------
template <class A, class B>
struct Cat2;
template <int... a, int... b>
struct Cat2<Strides<a...>, Strides<b...>>
{
    using type = Strides<a..., b...>;
};
template <class A, class... R>
struct CatN
{
    using type = typename Cat2<A, typename CatN<R...>::type>::type;
};
template <class A>
struct CatN<A>
{
    using type = A;
};
template <class... P>
using Cat = typename CatN<P...>::type;
template <int K>
struct Rep3;
template <>
struct Rep3<1> { using type = Strides<3>; };
template <>
struct Rep3<2> { using type = Strides<3, 3>; };
template <>
struct Rep3<3> { using type = Strides<3, 3, 3>; };

template <int K> using S_D_X1 = Cat<typename Rep3<K>::type, typename Rep3<K>::type, Strides<1>>;
template <int K> using S_D_X = typename Rep3<K>::type;
template <int K> using S_RB_X1 = Cat<Strides<1>, typename Rep3<K>::type, Strides<3, 3, 3, 4>>;
template <int K> using S_RB_X = typename Rep3<K>::type;
template <int K> using S_D_V1 = typename Rep3<K>::type;
template <int K> using S_RB_V1 = S_RB_X1<K>;
using S_D_EDGE = Cat<S_D_X1<2>, S_D_X<2>>;
using S_D_EDGE_POINT = Cat<S_D_X1<2>, S_D_X<2>, S_D_X1<1>>;
using S_RB_EDGE = Cat<S_RB_X1<2>, S_RB_X<2>>;
using S_RB_EDGE_POINT = Cat<S_RB_X1<2>, S_RB_X<2>, S_RB_X1<1>>;
using S_TAIL = Strides<1, 1, 1>;

// ---- cursor over the gathered inputs -------------------------------------------------------------------------------------
template <class T, int NS, int NR>
struct Cur
{
    const Loader<T>& L;
    int o = 0, is = 0, ir = 0;
    MS_HD explicit Cur(const Loader<T>& l) : L(l) {}
    MS_HD double s() { return L.s(o++); }
    MS_HD V3<double> v()
    {
        const V3<double> r = L.v(o);
        o += 3;
        return r;
    }
    MS_HD V3<T> soft_dof()
    {
        const V3<T> r = L.dof(o, is++);
        o += 3;
        return r;
    }
    template <int K>
    MS_HD void d_x1(V3<T>* x1)
    {
        V3<T> v1[K];
        for (int k = 0; k < K; k++) v1[k] = soft_dof();
        V3<double> x0[K];
        for (int k = 0; k < K; k++) x0[k] = v();
        const double dt = s();
        for (int k = 0; k < K; k++) x1[k] = x0[k] + dt * v1[k];
    }
    template <int K>
    MS_HD void d_v1(V3<T>* v1)
    {
        for (int k = 0; k < K; k++) v1[k] = soft_dof();
    }
    template <int K>
    MS_HD void rest(V3<double>* X)
    {
        for (int k = 0; k < K; k++) X[k] = v();
    }
    // positions (VEL = false) or velocities (VEL = true) of K points of one rigid body
    template <int K, bool VEL>
    MS_HD void rb(V3<T>* out)
    {
        const double dt = s();
        V3<double> xl[K];
        for (int k = 0; k < K; k++) xl[k] = v();
        const V3<T> v1 = L.dof(o, NS + ir);
        const V3<T> w1 = L.dof(o + 3, NS + NR + ir);
        const V3<double> t0 = L.v(o + 6);
        const M3<T> R1 = rb_R1(L.in + o + 9, w1, dt);
        o += 13;
        ir++;
        for (int k = 0; k < K; k++) {
            const V3<T> r = R1 * xl[k];
            if (VEL) out[k] = v1 + cross(w1, r);
            else out[k] = (t0 + dt * v1) + r;
        }
    }
};

// ---- distances (stark/src/models/distances.cpp:57-109) ------------------------------------------------------------------------
template <class T>
MS_HD T distance_point_point(const V3<T>& p, const V3<T>& q) { return sqrt(sqnorm(p - q)); }
template <class T>
MS_HD T distance_point_line(const V3<T>& p, const V3<T>& a, const V3<T>& b) { return sqrt(sq_distance_point_line(p, a, b)); }
template <class T>
MS_HD T distance_point_plane(const V3<T>& p, const V3<T>& a, const V3<T>& b, const V3<T>& c)
{
    const V3<T> n = normalized(cross(a - c, b - c));
    const T d = dot(p - a, n);
    return sqrt(d * d);
}
template <class T>
MS_HD T distance_line_line(const V3<T>& a, const V3<T>& b, const V3<T>& p, const V3<T>& q)
{
    const V3<T> n = cross(b - a, q - p);
    const T l = dot(p - a, n);
    return sqrt(l * l * inv(sqnorm(n)));
}
// cubic barrier (EnergyFrictionalContact.cpp:1225-1237) and edge-edge mollifier (:1251-1259)
template <class T>
MS_HD T barrier(const T& d, double dhat, double k) { return (k / 3.0) * pow3(dhat - d); }
template <class T>
MS_HD T ee_mollifier(const V3<T>* ea, const V3<T>* eb, const V3<double>* ea_rest, const V3<double>* eb_rest)
{
    const double eps_x = 1e-3 * sqnorm(ea_rest[0] - ea_rest[1]) * sqnorm(eb_rest[0] - eb_rest[1]);
    const T x = sqnorm(cross(ea[1] - ea[0], eb[1] - eb[0]));
    if (val(x) > eps_x) return T(1.0);
    const T r = x * (1.0 / eps_x);
    return (2.0 - r) * r;
}
// C0 friction (EnergyFrictionalContact.cpp:1260-1278, :1321-1329): tail bindings T (2x3), mu, fn, epsv, dt
template <class T, class C>
MS_HD T friction_tail(C& c, const V3<T>& v)
{
    double Tm[6];
    for (int i = 0; i < 6; i++) Tm[i] = c.s();
    const double mu = c.s(), fn = c.s(), epsv = c.s(), dt = c.s();
    const T ut0 = (Tm[0] * v.x + Tm[1] * v.y + Tm[2] * v.z) * dt + 1.13e-9;
    const T ut1 = (Tm[3] * v.x + Tm[4] * v.y + Tm[5] * v.z) * dt - 1.07e-9;
    const T u = sqrt(ut0 * ut0 + ut1 * ut1);
    const double epsu = dt * epsv;
    const double k = mu * fn / epsu;
    const double eps = mu * fn / (2.0 * k);
    if (val(u) < epsu) return (0.5 * k) * (u * u);
    return (mu * fn) * (u - eps);
}

// ---- barrier potentials ---------------------------------------------------------------------------------------------------------
enum Src { SD = 0, SRB = 1 };
// point-triangle family: first object contributes KA points, second KB points; DIST: 0 point-point, 1 point-line
// (point = first), 2 point-plane (point = first), 3 point-line (point = second), 4 point-plane (point = second)
template <Src A, int KA, Src B, int KB, int DIST>
struct PT_Contact
{
    static constexpr int NS = (A == SD ? KA : 0) + (B == SD ? KB : 0);
    static constexpr int NR = (A == SRB ? 1 : 0) + (B == SRB ? 1 : 0);
    static constexpr int NB = NS + 2 * NR;
    using PA = std::conditional_t<A == SD, S_D_X1<KA>, S_RB_X1<KA>>;
    using PB = std::conditional_t<B == SD, S_D_X1<KB>, S_RB_X1<KB>>;
    using Layout = Cat<PA, PB, S_TAIL>;
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        Cur<T, NS, NR> c(L);
        V3<T> a[KA], b[KB];
        if (A == SD) c.template d_x1<KA>(a);
        else c.template rb<KA, false>(a);
        if (B == SD) c.template d_x1<KB>(b);
        else c.template rb<KB, false>(b);
        const double dhat = c.s() + c.s(), k = c.s();  // _barrier_potential: dhat_a, dhat_b, k (:1307-1312)
        T d;
        if (DIST == 0) d = distance_point_point(a[0], b[0]);
        else if (DIST == 1) d = distance_point_line(a[0], b[0], b[KB > 1 ? 1 : 0]);
        else if (DIST == 2) d = distance_point_plane(a[0], b[0], b[KB > 1 ? 1 : 0], b[KB > 2 ? 2 : 0]);
        else if (DIST == 3) d = distance_point_line(b[0], a[0], a[KA > 1 ? 1 : 0]);
        else d = distance_point_plane(b[0], a[0], a[KA > 1 ? 1 : 0], a[KA > 2 ? 2 : 0]);
        return barrier(d, dhat, k);
    }
};
// edge-edge family. PA / PB: 1 = the object also binds a point (edge + point getter). DIST: 0 point-point (p,q),
// 1 point(a)-line(eb), 2 line-line, 3 point(b)-line(ea)
template <Src A, bool PA_, Src B, bool PB_, int DIST>
struct EE_Contact
{
    static constexpr int NS = (A == SD ? 2 + (PA_ ? 1 : 0) : 0) + (B == SD ? 2 + (PB_ ? 1 : 0) : 0);
    static constexpr int NR = (A == SRB ? 1 + (PA_ ? 1 : 0) : 0) + (B == SRB ? 1 + (PB_ ? 1 : 0) : 0);
    static constexpr int NB = NS + 2 * NR;
    using LA = std::conditional_t<A == SD, std::conditional_t<PA_, S_D_EDGE_POINT, S_D_EDGE>, std::conditional_t<PA_, S_RB_EDGE_POINT, S_RB_EDGE>>;
    using LB = std::conditional_t<B == SD, std::conditional_t<PB_, S_D_EDGE_POINT, S_D_EDGE>, std::conditional_t<PB_, S_RB_EDGE_POINT, S_RB_EDGE>>;
    using Layout = Cat<LA, LB, S_TAIL>;
    template <class T, Src S, bool P>
    MS_HD static void side(Cur<T, NS, NR>& c, V3<T>* e, V3<double>* rest, V3<T>* p)
    {
        if (S == SD) c.template d_x1<2>(e);
        else c.template rb<2, false>(e);
        c.template rest<2>(rest);
        if (P) {
            if (S == SD) c.template d_x1<1>(p);
            else c.template rb<1, false>(p);
        }
    }
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        Cur<T, NS, NR> c(L);
        V3<T> ea[2], eb[2], p[1], q[1];
        V3<double> ra[2], rb_[2];
        side<T, A, PA_>(c, ea, ra, p);
        side<T, B, PB_>(c, eb, rb_, q);
        const double k = c.s(), dhat = c.s() + c.s();  // _edge_edge_mollified_barrier_potential: k, dhat_a, dhat_b (:1313-1318)
        T d;
        if (DIST == 0) d = distance_point_point(p[0], q[0]);
        else if (DIST == 1) d = distance_point_line(p[0], eb[0], eb[1]);
        else if (DIST == 2) d = distance_line_line(ea[0], ea[1], eb[0], eb[1]);
        else d = distance_point_line(q[0], ea[0], ea[1]);
        return ee_mollifier(ea, eb, ra, rb_) * barrier(d, dhat, k);
    }
};

// ---- friction potentials --------------------------------------------------------------------------------------------------------------
// first object: KA points, second: KB points. KIND: 0 point-point, 1 point(a)-edge(b), 2 point(a)-triangle(b), 3 edge-edge,
// 4 edge(a)-point(b), 5 triangle(a)-point(b)
template <Src A, int KA, Src B, int KB, int KIND>
struct Friction
{
    static constexpr int NS = (A == SD ? KA : 0) + (B == SD ? KB : 0);
    static constexpr int NR = (A == SRB ? 1 : 0) + (B == SRB ? 1 : 0);
    static constexpr int NB = NS + 2 * NR;
    static constexpr int NBARY = KIND == 0 ? 0 : ((KIND == 2 || KIND == 5) ? 3 : 2);
    using PA = std::conditional_t<A == SD, S_D_V1<KA>, S_RB_V1<KA>>;
    using PB = std::conditional_t<B == SD, S_D_V1<KB>, S_RB_V1<KB>>;
    using SB = std::conditional_t<NBARY == 0, Strides<6, 1, 1, 1, 1>, std::conditional_t<NBARY == 2, Strides<2, 6, 1, 1, 1, 1>, Strides<3, 6, 1, 1, 1, 1>>>;
    using Layout = Cat<PA, PB, SB>;
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        Cur<T, NS, NR> c(L);
        V3<T> a[KA], b[KB];
        if (A == SD) c.template d_v1<KA>(a);
        else c.template rb<KA, true>(a);
        if (B == SD) c.template d_v1<KB>(b);
        else c.template rb<KB, true>(b);
        double bary[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < NBARY; i++) bary[i] = c.s();
        V3<T> v;
        if (KIND == 0) v = b[0] - a[0];                                                                       // vq - vp (:1083)
        else if (KIND == 1) v = (bary[0] * b[0] + bary[1] * b[KB > 1 ? 1 : 0]) - a[0];                       // _friction_point_edge (:1330-1338)
        else if (KIND == 2) v = (bary[0] * b[0] + bary[1] * b[KB > 1 ? 1 : 0] + bary[2] * b[KB > 2 ? 2 : 0]) - a[0];  // (:1339-1347)
        else if (KIND == 3) v = (b[0] + bary[1] * (b[KB > 1 ? 1 : 0] - b[0])) - (a[0] + bary[0] * (a[KA > 1 ? 1 : 0] - a[0]));  // (:1348-1356)
        else if (KIND == 4) v = (bary[0] * a[0] + bary[1] * a[KA > 1 ? 1 : 0]) - b[0];                       // D -> RB point-edge (:1196-1204)
        else v = (bary[0] * a[0] + bary[1] * a[KA > 1 ? 1 : 0] + bary[2] * a[KA > 2 ? 2 : 0]) - b[0];        // D -> RB point-triangle (:1206-1214)
        return friction_tail<T>(c, v);
    }
};

#define MS_POT(NAME, STR, ...)                         \
    struct NAME : __VA_ARGS__                          \
    {                                                  \
        static constexpr const char* name = STR;       \
    };
// barrier: deformable - deformable (EnergyFrictionalContact.cpp:833-897)
MS_POT(C_dd_pt_pp, "contact_d_d_pt_pp_cubic", PT_Contact<SD, 1, SD, 1, 0>)
MS_POT(C_dd_pt_pe, "contact_d_d_pt_pe_cubic", PT_Contact<SD, 1, SD, 2, 1>)
MS_POT(C_dd_pt_pt, "contact_d_d_pt_pt_cubic", PT_Contact<SD, 1, SD, 3, 2>)
MS_POT(C_dd_ee_pp, "contact_d_d_ee_pp_cubic", EE_Contact<SD, true, SD, true, 0>)
MS_POT(C_dd_ee_pe, "contact_d_d_ee_pe_cubic", EE_Contact<SD, true, SD, false, 1>)
MS_POT(C_dd_ee_ee, "contact_d_d_ee_ee_cubic", EE_Contact<SD, false, SD, false, 2>)
// barrier: rigid - rigid (:904-967)
MS_POT(C_rr_pt_pp, "contact_rb_rb_pt_pp_cubic", PT_Contact<SRB, 1, SRB, 1, 0>)
MS_POT(C_rr_pt_pe, "contact_rb_rb_pt_pe_cubic", PT_Contact<SRB, 1, SRB, 2, 1>)
MS_POT(C_rr_pt_pt, "contact_rb_rb_pt_pt_cubic", PT_Contact<SRB, 1, SRB, 3, 2>)
MS_POT(C_rr_ee_pp, "contact_rb_rb_ee_pp_cubic", EE_Contact<SRB, true, SRB, true, 0>)
MS_POT(C_rr_ee_pe, "contact_rb_rb_ee_pe_cubic", EE_Contact<SRB, true, SRB, false, 1>)
MS_POT(C_rr_ee_ee, "contact_rb_rb_ee_ee_cubic", EE_Contact<SRB, false, SRB, false, 2>)
// barrier: rigid - deformable (:974-1072)
MS_POT(C_rd_pt_pp, "contact_rb_d_pt_pp_cubic", PT_Contact<SRB, 1, SD, 1, 0>)
MS_POT(C_rd_pt_pe, "contact_rb_d_pt_pe_cubic", PT_Contact<SRB, 1, SD, 2, 1>)
MS_POT(C_rd_pt_pt, "contact_rb_d_pt_pt_cubic", PT_Contact<SRB, 1, SD, 3, 2>)
MS_POT(C_rd_pt_ep, "contact_rb_d_pt_ep_cubic", PT_Contact<SRB, 2, SD, 1, 3>)
MS_POT(C_rd_pt_tp, "contact_rb_d_pt_tp_cubic", PT_Contact<SRB, 3, SD, 1, 4>)
MS_POT(C_rd_ee_pp, "contact_rb_d_ee_pp_cubic", EE_Contact<SRB, true, SD, true, 0>)
MS_POT(C_rd_ee_pe, "contact_rb_d_ee_pe_cubic", EE_Contact<SRB, true, SD, false, 1>)
MS_POT(C_rd_ee_ee, "contact_rb_d_ee_ee_cubic", EE_Contact<SRB, false, SD, false, 2>)
MS_POT(C_rd_ee_ep, "contact_rb_d_ee_ep_cubic", EE_Contact<SRB, false, SD, true, 3>)
// friction (:1078-1116, :1121-1159, :1164-1218)
MS_POT(F_dd_pp, "friction_d_d_pp_C0", Friction<SD, 1, SD, 1, 0>)
MS_POT(F_dd_pe, "friction_d_d_pe_C0", Friction<SD, 1, SD, 2, 1>)
MS_POT(F_dd_pt, "friction_d_d_pt_C0", Friction<SD, 1, SD, 3, 2>)
MS_POT(F_dd_ee, "friction_d_d_ee_C0", Friction<SD, 2, SD, 2, 3>)
MS_POT(F_rr_pp, "friction_rb_rb_pp_C0", Friction<SRB, 1, SRB, 1, 0>)
MS_POT(F_rr_pe, "friction_rb_rb_pe_C0", Friction<SRB, 1, SRB, 2, 1>)
MS_POT(F_rr_pt, "friction_rb_rb_pt_C0", Friction<SRB, 1, SRB, 3, 2>)
MS_POT(F_rr_ee, "friction_rb_rb_ee_C0", Friction<SRB, 2, SRB, 2, 3>)
MS_POT(F_rd_pp, "friction_rb_d_pp_C0", Friction<SRB, 1, SD, 1, 0>)
MS_POT(F_rd_pe, "friction_rb_d_pe_C0", Friction<SRB, 1, SD, 2, 1>)
MS_POT(F_rd_pt, "friction_rb_d_pt_C0", Friction<SRB, 1, SD, 3, 2>)
MS_POT(F_rd_ee, "friction_rb_d_ee_C0", Friction<SRB, 2, SD, 2, 3>)
MS_POT(F_rd_ep, "friction_rb_d_ep_C0", Friction<SRB, 2, SD, 1, 4>)
MS_POT(F_rd_tp, "friction_rb_d_tp_C0", Friction<SRB, 3, SD, 1, 5>)
#undef MS_POT

#define MISTARK_FOR_EACH_CONTACT_ENERGY(X)                                                                                            \
    X(C_dd_pt_pp) X(C_dd_pt_pe) X(C_dd_pt_pt) X(C_dd_ee_pp) X(C_dd_ee_pe) X(C_dd_ee_ee)                                             \
    X(C_rr_pt_pp) X(C_rr_pt_pe) X(C_rr_pt_pt) X(C_rr_ee_pp) X(C_rr_ee_pe) X(C_rr_ee_ee)                                             \
    X(C_rd_pt_pp) X(C_rd_pt_pe) X(C_rd_pt_pt) X(C_rd_pt_ep) X(C_rd_pt_tp) X(C_rd_ee_pp) X(C_rd_ee_pe) X(C_rd_ee_ee) X(C_rd_ee_ep) \
    X(F_dd_pp) X(F_dd_pe) X(F_dd_pt) X(F_dd_ee) X(F_rr_pp) X(F_rr_pe) X(F_rr_pt) X(F_rr_ee)                                         \
    X(F_rd_pp) X(F_rd_pe) X(F_rd_pt) X(F_rd_ee) X(F_rd_ep) X(F_rd_tp)

}  // namespace mistark
