// energies.hpp — the element energies on the hot path, written once against a generic scalar T
// (T = double for energy-only evaluation, T = HDual for gradient + Hessian; see hdual.hpp).
//
// One struct per potential registry key of the reference. `Layout` lists the strides of the bound
// inputs in the reference's binding order (the order of the mws.make_* calls in the cited constructor), `NB` is the
// number of 3-DoF blocks and `dof_binding[k]` the binding that provides local DoF block k. Local DoF order follows the
// reference: DoF sets in registration order, then binding order
// (symx/src/solver/second_order/SecondOrderCompiledPotential.cpp:10-33).
// All DoFs are next-step velocities: x1 = x0 + dt v1 (stark/src/models/time_integration.cpp:3-11).
#pragma once
#include <type_traits>

#include "hdual.hpp"

namespace mistark {

template <int... S>
struct Strides
{
    static constexpr int NBIND = sizeof...(S);
    static constexpr int NIN = (S + ...);
    // f(binding index, stride, offset into the gathered input array)
    template <class F>
    MS_HD static void for_each(F&& f)
    {
        int o = 0, b = 0;
        ((f(b, S, o), o += S, b++), ...);
    }
    static void strides(int* out)
    {
        int b = 0;
        ((out[b++] = S), ...);
    }
};

// Typed view of the gathered inputs of one element. (si, sj): local DoF indices seeded with eps1 / eps2.
template <class T>
struct Loader;
template <>
struct Loader<double>
{
    const double* in;
    MS_HD double s(int o) const { return in[o]; }
    MS_HD V3<double> v(int o) const { return {in[o], in[o + 1], in[o + 2]}; }
    MS_HD V3<double> dof(int o, int) const { return {in[o], in[o + 1], in[o + 2]}; }
};
template <>
struct Loader<HDual>
{
    const double* in;
    int si, sj;
    MS_HD double s(int o) const { return in[o]; }
    MS_HD V3<double> v(int o) const { return {in[o], in[o + 1], in[o + 2]}; }
    MS_HD HDual dofc(int o, int l) const { return HDual(in[o], l == si ? 1.0 : 0.0, l == sj ? 1.0 : 0.0, 0.0); }
    MS_HD V3<HDual> dof(int o, int k) const { return {dofc(o, 3 * k), dofc(o + 1, 3 * k + 1), dofc(o + 2, 3 * k + 2)}; }
};

// ======================================================================================================================
// stark/src/models/deformables/point/EnergyLumpedInertia.cpp:12-49
// bindings: v1*, x0, v0, a, f, lumped_volume[idx], density[group], damping[group], is_quasistatic[group], dt, gravity
struct E_LumpedInertia
{
    static constexpr const char* name = "EnergyLumpedInertia";
    using Layout = Strides<3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 3>;
    static constexpr int NB = 1;
    static constexpr int dof_binding[NB] = {0};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const V3<T> v1 = L.dof(0, 0);
        const V3<double> x0 = L.v(3), v0 = L.v(6), a = L.v(9), f = L.v(12);
        const double volume = L.s(15), density = L.s(16), damping = L.s(17), is_quasistatic = L.s(18), dt = L.s(19);
        const V3<double> gravity = L.v(20);
        const double mass = volume * density;
        const V3<T> x1 = x0 + dt * v1;
        const V3<double> xhat = x0 + dt * v0;
        const V3<T> dev = x1 - xhat;
        const V3<T> dev2 = x1 - x0;
        const T E_inertia = 0.5 * mass * (dot(dev, dev) * (1.0 / (dt * dt)) + dot(dev2, dev2) * (damping / dt));
        const V3<double> f_ext = mass * (a + gravity) + f;
        const T E_ext = -dot(f_ext, x1);
        return E_ext + select(is_quasistatic > 0.5, T(0.0), E_inertia);
    }
};

// stark/src/models/deformables/point/EnergyPrescribedPositions.cpp:15-32
// bindings: v1*, x0, target[idx], stiffness[group], dt
struct E_PrescribedPositions
{
    static constexpr const char* name = "EnergyPrescribedPositions";
    using Layout = Strides<3, 3, 3, 1, 1>;
    static constexpr int NB = 1;
    static constexpr int dof_binding[NB] = {0};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const V3<T> v1 = L.dof(0, 0);
        const V3<double> x0 = L.v(3), target = L.v(6);
        const double k = L.s(9), dt = L.s(10);
        const V3<T> x1 = x0 + dt * v1;
        return 0.5 * k * sqnorm(x1 - target);
    }
};

// ----------------------------------------------------------------------------------------------------------------------
// Stable Neo-Hookean density, stark/src/models/deformables/volume/EnergyTetStrain.cpp:50-61
template <class T>
MS_HD T stable_neohookean_density(const M3<T>& F, double e, double nu)
{
    const double mu = e / (2.0 * (1.0 + nu));
    const double lambda = (e * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const double mu_ = 4.0 / 3.0 * mu;
    const double lambda_ = lambda + 5.0 / 6.0 * mu;
    const T detF = det(F);
    const T Ic = frob_sq(F);
    const double alpha = 1.0 + mu_ / lambda_ - mu_ / (4.0 * lambda_);
    return 0.5 * mu_ * (Ic - 3.0) + 0.5 * lambda_ * pow2(detF - alpha) - 0.5 * mu_ * log(Ic + 1.0);
}
template <class T>
MS_HD M3<T> green_strain(const M3<T>& F)
{
    M3<T> E = transpose(F) * F;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) E.m[i][j] = 0.5 * (E.m[i][j] - (i == j ? 1.0 : 0.0));
    return E;
}

// stark/src/models/deformables/volume/EnergyTetStrain.cpp:80-123
// bindings: v1[4]*, x0[4], X[4], scale, youngs_modulus, poissons_ratio (all [group]), dt
struct E_TetStrainEO
{
    static constexpr const char* name = "EnergyTetStrain_Elasticity_Only";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double scale = L.s(36), e = L.s(37), nu = L.s(38), dt = L.s(39);
        V3<T> x1[4];
        V3<double> Xs[4];
        for (int i = 0; i < 4; i++) {
            x1[i] = L.v(12 + 3 * i) + dt * L.dof(3 * i, i);
            Xs[i] = scale * L.v(24 + 3 * i);
        }
        const M3<double> DX = from_cols(Xs[1] - Xs[0], Xs[2] - Xs[0], Xs[3] - Xs[0]);
        const M3<double> DXinv = inverse(DX);
        const M3<T> Dx1 = from_cols(x1[1] - x1[0], x1[2] - x1[0], x1[3] - x1[0]);
        const M3<T> F1 = Dx1 * DXinv;
        const double vol = det(DX) / 6.0;
        return vol * stable_neohookean_density(F1, e, nu);
    }
};

// stark/src/models/deformables/volume/EnergyTetStrain.cpp:12-78
// bindings: v1[4]*, x0[4], X[4], scale, e, nu, strain_limit, strain_limit_stiffness, damping (all [group]), dt
struct E_TetStrain
{
    static constexpr const char* name = "EnergyTetStrain";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double scale = L.s(36), e = L.s(37), nu = L.s(38), strain_limit = L.s(39), sl_k = L.s(40), damping = L.s(41), dt = L.s(42);
        V3<T> x1[4];
        V3<double> x0[4], Xs[4];
        for (int i = 0; i < 4; i++) {
            x0[i] = L.v(12 + 3 * i);
            x1[i] = x0[i] + dt * L.dof(3 * i, i);
            Xs[i] = scale * L.v(24 + 3 * i);
        }
        const M3<double> DX = from_cols(Xs[1] - Xs[0], Xs[2] - Xs[0], Xs[3] - Xs[0]);
        const M3<double> DXinv = inverse(DX);
        const M3<T> F1 = from_cols(x1[1] - x1[0], x1[2] - x1[0], x1[3] - x1[0]) * DXinv;
        const M3<T> E1 = green_strain(F1);
        const double vol = det(DX) / 6.0;
        const M3<double> F0 = from_cols(x0[1] - x0[0], x0[2] - x0[0], x0[3] - x0[0]) * DXinv;
        const M3<double> E0 = green_strain(F0);
        T dsum = T(0.0);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) dsum = dsum + pow2((E1.m[i][j] - E0.m[i][j]) * (1.0 / dt));
        const T elastic = stable_neohookean_density(F1, e, nu);
        const T damp = 0.5 * damping * dsum;
        // smooth upper-bound proxy of the largest eigenvalue of E (EnergyTetStrain.cpp:66-71)
        const T trE = trace(E1);
        M3<T> devE = E1;
        for (int i = 0; i < 3; i++) devE.m[i][i] = devE.m[i][i] - trE * (1.0 / 3.0);
        const T dev_sq = frob_sq(devE);
        const double largest_v = val(trE) / 3.0 + ::sqrt(2.0 / 3.0) * ::sqrt(val(dev_sq));
        T sl = T(0.0);
        if (largest_v - strain_limit > 0.0) {
            const T dl = trE * (1.0 / 3.0) + ::sqrt(2.0 / 3.0) * sqrt(dev_sq) - strain_limit;
            sl = (sl_k / 3.0) * pow3(dl);
        }
        return vol * (elastic + damp + sl);
    }
};

// ----------------------------------------------------------------------------------------------------------------------
// Triangle membrane. stark/src/models/deformables/surface/EnergyTriangleStrain.cpp:13-80 (full), :82-129 (elasticity only)
// rest Jacobian: deformable_tools.cpp:7-21; eigenvalues_sym_2x2: deformable_tools.cpp:26-36
// bindings (full): v1[3]*, x0[3], X[3], scale, thickness, e, nu, strain_damping, strain_limit, strain_limit_stiffness, inflation, dt
// bindings (EO)  : v1[3]*, x0[3], X[3], scale, thickness, e, nu, inflation, dt
template <bool FULL>
struct E_TriangleStrainT
{
    static constexpr int NB = 3;
    static constexpr int dof_binding[NB] = {0, 1, 2};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const int p = 27;
        const double scale = L.s(p), thickness = L.s(p + 1), e = L.s(p + 2), nu = L.s(p + 3);
        const double damping = FULL ? L.s(p + 4) : 0.0, strain_limit = FULL ? L.s(p + 5) : 0.0, sl_k = FULL ? L.s(p + 6) : 0.0;
        const double inflation = L.s(FULL ? p + 7 : p + 4), dt = L.s(FULL ? p + 8 : p + 5);
        V3<T> x1[3];
        V3<double> x0[3], Xs[3];
        for (int i = 0; i < 3; i++) {
            x0[i] = L.v(9 + 3 * i);
            x1[i] = x0[i] + dt * L.dof(3 * i, i);
            Xs[i] = scale * L.v(18 + 3 * i);
        }
        const double rest_area = 0.5 * norm(cross(Xs[0] - Xs[2], Xs[1] - Xs[2]));
        // rest configuration projected into its own plane -> 2x2 Jacobian and its inverse
        const V3<double> u = normalized(Xs[1] - Xs[0]);
        const V3<double> n = cross(u, Xs[2] - Xs[0]);
        const V3<double> v = normalized(cross(u, n));
        const double a00 = dot(u, Xs[1]) - dot(u, Xs[0]), a01 = dot(u, Xs[2]) - dot(u, Xs[0]);
        const double a10 = dot(v, Xs[1]) - dot(v, Xs[0]), a11 = dot(v, Xs[2]) - dot(v, Xs[0]);
        const double idet = 1.0 / (a00 * a11 - a01 * a10);
        const double i00 = a11 * idet, i01 = -a01 * idet, i10 = -a10 * idet, i11 = a00 * idet;
        // F (3x2) = [x1-x0 | x2-x0] * DXinv
        const V3<T> d1 = x1[1] - x1[0], d2 = x1[2] - x1[0];
        const V3<T> f0 = i00 * d1 + i10 * d2, f1 = i01 * d1 + i11 * d2;
        const T C00 = dot(f0, f0), C01 = dot(f0, f1), C11 = dot(f1, f1);
        const double mu = e / (2.0 * (1.0 + nu));
        const double lambda = (e * nu) / ((1.0 + nu) * (1.0 - nu));  // 2D
        const T area = 0.5 * norm(cross(x1[0] - x1[2], x1[1] - x1[2]));
        const T J = area * (1.0 / rest_area);
        const T Ic = C00 + C11;
        const T logJ = log(J);
        T density = 0.5 * mu * (Ic - 2.0) - mu * logJ + 0.5 * lambda * pow2(logJ);
        const V3<double> n0 = -normalized(cross(x0[1] - x0[0], x0[2] - x0[0]));
        density = density + (inflation / 3.0) * dot(n0, x1[0] + x1[1] + x1[2]);
        if (FULL) {
            const T E00 = 0.5 * (C00 - 1.0), E01 = 0.5 * C01, E11 = 0.5 * (C11 - 1.0);
            const V3<double> e1 = x0[1] - x0[0], e2 = x0[2] - x0[0];
            const V3<double> g0 = i00 * e1 + i10 * e2, g1 = i01 * e1 + i11 * e2;
            const double P00 = 0.5 * (dot(g0, g0) - 1.0), P01 = 0.5 * dot(g0, g1), P11 = 0.5 * (dot(g1, g1) - 1.0);
            const double idt = 1.0 / dt;
            density = density + 0.5 * damping * (pow2((E00 - P00) * idt) + 2.0 * pow2((E01 - P01) * idt) + pow2((E11 - P11) * idt));
            // strain limiting on both eigenvalues of E
            const T disc = 4.0 * pow2(E01) + pow2(E00 - E11);
            const double sq = ::sqrt(val(disc));
            const double s0 = 0.5 * (val(E00) + val(E11) + sq), s1 = 0.5 * (val(E00) + val(E11) - sq);
            if (s0 - strain_limit > 0.0) density = density + (sl_k / 3.0) * pow3(0.5 * (E00 + E11 + sqrt(disc)) - strain_limit);
            if (s1 - strain_limit > 0.0) density = density + (sl_k / 3.0) * pow3(0.5 * (E00 + E11 - sqrt(disc)) - strain_limit);
        }
        return (thickness * rest_area) * density;
    }
};
struct E_TriangleStrain : E_TriangleStrainT<true>
{
    static constexpr const char* name = "EnergyTriangleStrain";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1>;
};
struct E_TriangleStrainEO : E_TriangleStrainT<false>
{
    static constexpr const char* name = "EnergyTriangleStrain_Elasticity_Only";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1>;
};

// ----------------------------------------------------------------------------------------------------------------------
// stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:12-23
template <class T>
MS_HD T dihedral_angle(const V3<T>* x)
{
    const V3<T> e0 = x[1] - x[0], e1 = x[2] - x[0], e2 = x[3] - x[0];
    const V3<T> n0 = cross(e0, e1);
    const V3<T> n1 = -cross(e0, e2);
    return acos((1.0 - 1e-12) * dot(normalized(n0), normalized(n1)));
}
// stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:26-62
// bindings: v1[4]*, x0[4], rest_dihedral_angle[idx], rest_edge_length[idx], rest_height[idx], scale, stiffness, damping ([group]), dt
struct E_DiscreteShells
{
    static constexpr const char* name = "EnergyDiscreteShells";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double rest_angle = L.s(24), rest_len = L.s(25), rest_h = L.s(26), scale = L.s(27), k = L.s(28), damping = L.s(29), dt = L.s(30);
        V3<T> x1[4];
        V3<double> x0[4];
        for (int i = 0; i < 4; i++) {
            x0[i] = L.v(12 + 3 * i);
            x1[i] = x0[i] + dt * L.dof(3 * i, i);
        }
        const double ratio = (rest_len * scale) / (rest_h * scale);
        const T da1 = dihedral_angle(x1);
        const T dd = da1 - rest_angle;
        const double da0 = dihedral_angle(x0);
        return k * (dd * dd) * ratio + (damping / dt) * (0.5 * pow2(da1) - da0 * da1) * ratio;
    }
};
// stark/src/models/deformables/surface/EnergyDiscreteShells.cpp:64-92
// bindings: v1[4]*, x0[4], bergou_K[idx] (4), bergou_coef[idx], stiffness[group], dt
struct E_BendingFlat
{
    static constexpr const char* name = "EnergyBendingFlat";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 4, 1, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double coef = L.s(28), k = L.s(29), dt = L.s(30);
        V3<T> s(T(0.0), T(0.0), T(0.0));
        for (int i = 0; i < 4; i++) {
            const V3<T> x1 = L.v(12 + 3 * i) + dt * L.dof(3 * i, i);
            s = s + L.s(24 + i) * x1;
        }
        return (0.5 * k * coef) * sqnorm(s);
    }
};


// ======================================================================================================================
// Rigid bodies. Kinematics: stark/src/models/rigidbodies/rigidbody_transformations.cpp:54-160
//   q1 = normalize(q0 + dt/2 (0,w1) * q0),  x1 = t0 + dt v1 + R(q1) x_loc,  d1 = R(q1) d_loc      (quaternions are (w,x,y,z))
// Conditional potentials (is_active > 0, EnergyRigidBodyConstraints.cpp:43) expose `active()`.
// ======================================================================================================================
template <class T>
MS_HD M3<T> quat_to_rotation(const T& qw, const T& qx, const T& qy, const T& qz)
{
    const T tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const T txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    M3<T> R;
    R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
    return R;
}
// R(q1) for the integrated quaternion; q0 = (e,f,g,h) plain doubles at in[o..o+3]
template <class T>
MS_HD M3<T> rb_R1(const double* q0, const V3<T>& w, double dt)
{
    const double e = q0[0], f = q0[1], g = q0[2], h = q0[3];
    // (0,w) * q0  (quat_dot_product, rigidbody_transformations.cpp:91-112)
    const T p0 = -(w.x * f) - w.y * g - w.z * h;
    const T p1 = w.x * e + w.y * h - w.z * g;
    const T p2 = -(w.x * h) + w.y * e + w.z * f;
    const T p3 = w.x * g - w.y * f + w.z * e;
    const T q0n = e + (0.5 * dt) * p0, q1n = f + (0.5 * dt) * p1, q2n = g + (0.5 * dt) * p2, q3n = h + (0.5 * dt) * p3;
    const T r = inv(sqrt(q0n * q0n + q1n * q1n + q2n * q2n + q3n * q3n));
    return quat_to_rotation(q0n * r, q1n * r, q2n * r, q3n * r);
}
MS_HD M3<double> rb_R0(const double* q0) { return quat_to_rotation(q0[0], q0[1], q0[2], q0[3]); }
// bound block of one body as created by RigidBodyDynamics::get_x1: v1*, w1*, t0, q0_ at offsets o, o+3, o+6, o+9 (13 doubles)
template <class T>
MS_HD V3<T> rb_point(const Loader<T>& L, int o, int kv, int kw, const V3<double>& x_loc, double dt)
{
    const V3<T> v1 = L.dof(o, kv), w1 = L.dof(o + 3, kw);
    const M3<T> R1 = rb_R1(L.in + o + 9, w1, dt);
    return (L.v(o + 6) + dt * v1) + R1 * x_loc;
}
template <class T>
MS_HD V3<T> rb_dir(const Loader<T>& L, int ow, int oq, int kw, const V3<double>& d_loc, double dt)
{
    return rb_R1(L.in + oq, L.dof(ow, kw), dt) * d_loc;
}
MS_HD V3<double> rb_point0(const double* in, int o, const V3<double>& x_loc)
{
    const M3<double> R0 = rb_R0(in + o + 9);
    return V3<double>(in[o + 6], in[o + 7], in[o + 8]) + R0 * x_loc;
}

// stark/src/models/rigidbodies/EnergyRigidBodyInertia.cpp:13-39
// bindings: v1*, v0, a, force, mass, linear_damping, is_quasistatic (all [rb]), dt, gravity
struct E_RBInertiaLinear
{
    static constexpr const char* name = "EnergyRigidBodyInertia_Linear";
    using Layout = Strides<3, 3, 3, 3, 1, 1, 1, 1, 3>;
    static constexpr int NB = 1;
    static constexpr int dof_binding[NB] = {0};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const V3<T> v1 = L.dof(0, 0);
        const V3<double> v0 = L.v(3), a = L.v(6), f = L.v(9), gravity = L.v(16);
        const double m = L.s(12), damping = L.s(13), is_q = L.s(14), dt = L.s(15);
        const V3<T> dev = v1 - v0;
        const T E_inertia = 0.5 * m * dot(dev, dev) + (0.5 * m * damping * dt) * dot(v1, v1);
        const V3<double> f_ext = m * (a + gravity) + f;
        const T E_ext = (-dt) * dot(f_ext, v1);
        return E_ext + select(is_q > 0.5, T(0.0), E_inertia);
    }
};
// stark/src/models/rigidbodies/EnergyRigidBodyInertia.cpp:42-67
// bindings: w1*, w0, aa, torque, J0_glob (9), angular_damping, is_quasistatic, dt
struct E_RBInertiaAngular
{
    static constexpr const char* name = "EnergyRigidBodyInertia_Angular";
    using Layout = Strides<3, 3, 3, 3, 9, 1, 1, 1>;
    static constexpr int NB = 1;
    static constexpr int dof_binding[NB] = {0};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const V3<T> w1 = L.dof(0, 0);
        const V3<double> w0 = L.v(3), aa = L.v(6), t = L.v(9);
        M3<double> J;
        for (int i = 0; i < 9; i++) J.m[i / 3][i % 3] = L.s(12 + i);
        const double damping = L.s(21), is_q = L.s(22), dt = L.s(23);
        const V3<T> dev = w1 - w0;
        const T E_inertia = 0.5 * (dot(dev, J * dev) + dot(w1, J * w1) * (damping * dt));
        const V3<double> t_ext = J * aa + t;
        const T E_ext = (-dt) * dot(t_ext, w1);
        return E_ext + select(is_q > 0.5, T(0.0), E_inertia);
    }
};

#define MISTARK_ACTIVE_AT(OFF) \
    static constexpr bool COND = true; \
    MS_HD static bool active(const double* in) { return in[OFF] > 0.0; }

// EnergyRigidBodyConstraints.cpp:30-45. bindings: loc, target_glob, stiffness, is_active ([idx]), dt, {v1*, w1*, t0, q0_}[rb]
struct E_RBGlobalPoints
{
    static constexpr const char* name = "rb_constraint_global_points";
    using Layout = Strides<3, 3, 1, 1, 1, 3, 3, 3, 4>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {5, 6};
    MISTARK_ACTIVE_AT(7)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(6), dt = L.s(8);
        const V3<T> p = rb_point(L, 9, 0, 1, L.v(0), dt);
        return 0.5 * k * sqnorm(L.v(3) - p);
    }
};
// EnergyRigidBodyConstraints.cpp:47-62. bindings: d_loc, target_d_glob, stiffness, is_active, dt, w1*[rb], q0_[rb]
struct E_RBGlobalDirections
{
    static constexpr const char* name = "rb_constraint_global_directions";
    using Layout = Strides<3, 3, 1, 1, 1, 3, 4>;
    static constexpr int NB = 1;
    static constexpr int dof_binding[NB] = {5};
    MISTARK_ACTIVE_AT(7)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(6), dt = L.s(8);
        const V3<T> d = rb_dir(L, 9, 12, 0, L.v(0), dt);
        return 0.5 * k * sqnorm(L.v(3) - d);
    }
};
// Two-body constraints: body blocks {v1*, w1*, t0, q0_} of a then b. Local DoF order (sets first): v1_a, v1_b, w1_a, w1_b.
// EnergyRigidBodyConstraints.cpp:64-80. bindings: a_loc, b_loc, stiffness, is_active, dt, body a, body b
struct E_RBPoints
{
    static constexpr const char* name = "rb_constraint_points";
    using Layout = Strides<3, 3, 1, 1, 1, 3, 3, 3, 4, 3, 3, 3, 4>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {5, 9, 6, 10};
    MISTARK_ACTIVE_AT(7)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(6), dt = L.s(8);
        const V3<T> a1 = rb_point(L, 9, 0, 2, L.v(0), dt);
        const V3<T> b1 = rb_point(L, 22, 1, 3, L.v(3), dt);
        return 0.5 * k * sqnorm(b1 - a1);
    }
};
template <class T>
MS_HD T sq_distance_point_line(const V3<T>& p, const V3<T>& a, const V3<T>& b)
{
    // stark/src/models/distances.cpp:61-68
    const V3<T> ab = b - a, ap = p - a;
    const T e = dot(ap, ab);
    return dot(ap, ap) - e * e * inv(dot(ab, ab));
}
// EnergyRigidBodyConstraints.cpp:82-99. bindings: a_loc, da_loc, b_loc, stiffness, is_active, dt, body a, body b
struct E_RBPointOnAxis
{
    static constexpr const char* name = "rb_constraint_point_on_axis";
    using Layout = Strides<3, 3, 3, 1, 1, 1, 3, 3, 3, 4, 3, 3, 3, 4>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {6, 10, 7, 11};
    MISTARK_ACTIVE_AT(10)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(9), dt = L.s(11);
        const V3<T> a1 = rb_point(L, 12, 0, 2, L.v(0), dt);
        const V3<T> da1 = rb_dir(L, 15, 21, 2, L.v(3), dt);
        const V3<T> b1 = rb_point(L, 25, 1, 3, L.v(6), dt);
        return 0.5 * k * sq_distance_point_line(b1, a1, a1 + da1);
    }
};
// EnergyRigidBodyConstraints.cpp:101-118. bindings: a_loc, b_loc, target_distance, stiffness, is_active, dt, body a, body b
struct E_RBDistances
{
    static constexpr const char* name = "rb_constraint_distances";
    using Layout = Strides<3, 3, 1, 1, 1, 1, 3, 3, 3, 4, 3, 3, 3, 4>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {6, 10, 7, 11};
    MISTARK_ACTIVE_AT(8)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double target = L.s(6), k = L.s(7), dt = L.s(9);
        const V3<T> a1 = rb_point(L, 10, 0, 2, L.v(0), dt);
        const V3<T> b1 = rb_point(L, 23, 1, 3, L.v(3), dt);
        return 0.5 * k * pow2(target - norm(b1 - a1));
    }
};
// EnergyRigidBodyConstraints.cpp:120-138. bindings: a_loc, b_loc, min_distance, max_distance, stiffness, is_active, dt, body a, body b
struct E_RBDistanceLimits
{
    static constexpr const char* name = "rb_constraint_distance_limits";
    using Layout = Strides<3, 3, 1, 1, 1, 1, 1, 3, 3, 3, 4, 3, 3, 3, 4>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {7, 11, 8, 12};
    MISTARK_ACTIVE_AT(9)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double dmin = L.s(6), dmax = L.s(7), k = L.s(8), dt = L.s(10);
        const V3<T> a1 = rb_point(L, 11, 0, 2, L.v(0), dt);
        const V3<T> b1 = rb_point(L, 24, 1, 3, L.v(3), dt);
        const T length = norm(b1 - a1);
        T E = T(0.0);
        if (val(length) < dmin) E = E + (0.5 * k) * pow2(dmin - length);
        if (val(length) > dmax) E = E + (0.5 * k) * pow2(length - dmax);
        return E;
    }
};
// EnergyRigidBodyConstraints.cpp:140-156. bindings: da_loc, db_loc, stiffness, is_active, dt, w1_a*, q0_a, w1_b*, q0_b
struct E_RBDirections
{
    static constexpr const char* name = "rb_constraint_directions";
    using Layout = Strides<3, 3, 1, 1, 1, 3, 4, 3, 4>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {5, 7};
    MISTARK_ACTIVE_AT(7)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(6), dt = L.s(8);
        const V3<T> da = rb_dir(L, 9, 12, 0, L.v(0), dt);
        const V3<T> db = rb_dir(L, 16, 19, 1, L.v(3), dt);
        return 0.5 * k * sqnorm(db - da);
    }
};
// EnergyRigidBodyConstraints.cpp:158-175. bindings: da_loc, db_loc, max_distance, stiffness, is_active, dt, w1_a*, q0_a, w1_b*, q0_b
struct E_RBAngleLimits
{
    static constexpr const char* name = "rb_constraint_angle_limits";
    using Layout = Strides<3, 3, 1, 1, 1, 1, 3, 4, 3, 4>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {6, 8};
    MISTARK_ACTIVE_AT(8)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double maxd = L.s(6), k = L.s(7), dt = L.s(9);
        const V3<T> da = rb_dir(L, 10, 13, 0, L.v(0), dt);
        const V3<T> db = rb_dir(L, 17, 20, 1, L.v(3), dt);
        const T length = norm(db - da);
        if (val(length) > maxd) return (k / 3.0) * pow3(length - maxd);
        return T(0.0);
    }
};
// EnergyRigidBodyConstraints.cpp:177-196. bindings: a_loc, b_loc, rest_length, stiffness, damping, is_active, dt, body a, body b
struct E_RBDampedSpring
{
    static constexpr const char* name = "rb_constraint_damped_spring";
    using Layout = Strides<3, 3, 1, 1, 1, 1, 1, 3, 3, 3, 4, 3, 3, 3, 4>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {7, 11, 8, 12};
    MISTARK_ACTIVE_AT(9)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double rest = L.s(6), k = L.s(7), damping = L.s(8), dt = L.s(10);
        const V3<T> a1 = rb_point(L, 11, 0, 2, L.v(0), dt);
        const V3<T> b1 = rb_point(L, 24, 1, 3, L.v(3), dt);
        const V3<double> a0 = rb_point0(L.in, 11, L.v(0)), b0 = rb_point0(L.in, 24, L.v(3));
        const T l1 = norm(b1 - a1);
        const double l0 = norm(b0 - a0);
        return 0.5 * k * pow2(l1 - rest) + 0.5 * damping * pow2((l1 - l0) * (1.0 / dt));
    }
};
// RigidBodyConstraints.h:54-69 (C1 controller, analogous to C1 friction)
template <class T>
MS_HD T c1_controller_energy(const V3<T>& da1, const V3<T>& va1, const V3<T>& vb1, double target, double max_force, double delay, double dt)
{
    const T v = dot(da1, vb1 - va1);
    const double k = max_force / delay, eps = delay / 2.0;
    const T dv = v - target;
    if (val(dv) < -delay) return (-max_force * dt) * (dv - eps);
    if (val(dv) < delay) return (0.5 * k * dt) * pow2(dv);
    return (max_force * dt) * (dv - eps);
}
// EnergyRigidBodyConstraints.cpp:198-218. bindings: da_loc, target_v, max_force, delay, is_active, v1_a*, v1_b*, w1_a*, q0_a, dt
struct E_RBLinearVelocity
{
    static constexpr const char* name = "rb_constraint_linear_velocity";
    using Layout = Strides<3, 1, 1, 1, 1, 3, 3, 3, 4, 1>;
    static constexpr int NB = 3;
    static constexpr int dof_binding[NB] = {5, 6, 7};
    MISTARK_ACTIVE_AT(6)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double target = L.s(3), max_force = L.s(4), delay = L.s(5), dt = L.s(20);
        const V3<T> va1 = L.dof(7, 0), vb1 = L.dof(10, 1);
        const V3<T> da1 = rb_dir(L, 13, 16, 2, L.v(0), dt);
        return c1_controller_energy(da1, va1, vb1, target, max_force, delay, dt);
    }
};
// EnergyRigidBodyConstraints.cpp:220-238. bindings: da_loc, target_w, max_torque, delay, is_active, w1_a*, w1_b*, q0_a, dt
struct E_RBAngularVelocity
{
    static constexpr const char* name = "rb_constraint_angular_velocity";
    using Layout = Strides<3, 1, 1, 1, 1, 3, 3, 4, 1>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {5, 6};
    MISTARK_ACTIVE_AT(6)
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double target = L.s(3), max_torque = L.s(4), delay = L.s(5), dt = L.s(17);
        const V3<T> wa1 = L.dof(7, 0), wb1 = L.dof(10, 1);
        const V3<T> da1 = rb_dir(L, 7, 13, 0, L.v(0), dt);
        return c1_controller_energy(da1, wa1, wb1, target, max_torque, delay, dt);
    }
};

// ======================================================================================================================
// Rods: stark/src/models/deformables/line/EnergySegmentStrain.cpp:11-55 (complete) and :57-88 (elasticity only)
// bindings: v1[2]*, x0[2], X[2], scale, section_radius, youngs_modulus, [strain_damping, strain_limit, strain_limit_stiffness,] dt
template <bool FULL>
struct E_SegmentStrainT
{
    using Layout = std::conditional_t<FULL, Strides<3, 3, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1>, Strides<3, 3, 3, 3, 3, 3, 1, 1, 1, 1>>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {0, 1};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double scale = L.s(18), radius = L.s(19), youngs_modulus = L.s(20);
        const double damping = FULL ? L.s(21) : 0.0, strain_limit = FULL ? L.s(22) : 0.0, sl_k = FULL ? L.s(23) : 0.0;
        const double dt = L.s(FULL ? 24 : 21);
        const V3<double> x00 = L.v(6), x01 = L.v(9);
        const V3<T> x10 = x00 + dt * L.dof(0, 0), x11 = x01 + dt * L.dof(3, 1);
        const double l_rest = norm(scale * L.v(12) - scale * L.v(15));
        const T l = norm(x10 - x11);
        const T e = (l - l_rest) * (1.0 / l_rest);
        const double volume = M_PI * radius * radius * l_rest;
        T E = (0.5 * volume * youngs_modulus) * pow2(e);
        if constexpr (FULL) {
            const T over = e - strain_limit;
            if (val(over) > 0.0) E = E + (volume * sl_k / 3.0) * pow3(over);
            const double e0 = (norm(x01 - x00) - l_rest) / l_rest;
            E = E + (0.5 * dt * damping) * pow2((e - e0) * (1.0 / dt));
        }
        return E;
    }
};
struct E_SegmentStrain : E_SegmentStrainT<true>
{
    static constexpr const char* name = "EnergySegmentStrain";
};
struct E_SegmentStrainEO : E_SegmentStrainT<false>
{
    static constexpr const char* name = "EnergySegmentStrain_Elasticity_Only";
};

// ======================================================================================================================
// Attachments (penalty springs between material points): stark/src/models/interactions/EnergyAttachments.cpp
// E = k/2 |q - p|^2 with p, q barycentric combinations of end-of-step positions.
// :17-35   d_d_p_p   bindings: v1[a,b]*, x0[a,b], k[group], dt
struct E_AttachPP
{
    static constexpr const char* name = "EnergyAttachments_d_d_p_p";
    using Layout = Strides<3, 3, 3, 3, 1, 1>;
    static constexpr int NB = 2;
    static constexpr int dof_binding[NB] = {0, 1};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(12), dt = L.s(13);
        const V3<T> a = L.v(6) + dt * L.dof(0, 0), b = L.v(9) + dt * L.dof(3, 1);
        return 0.5 * k * sqnorm(b - a);
    }
};
// :37-60   d_d_p_e   bindings: v1[p,e0,e1]*, x0[p,e0,e1], bary[idx] (2), k[group], dt
struct E_AttachPE
{
    static constexpr const char* name = "EnergyAttachments_d_d_p_e";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 2, 1, 1>;
    static constexpr int NB = 3;
    static constexpr int dof_binding[NB] = {0, 1, 2};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double b0 = L.s(18), b1 = L.s(19), k = L.s(20), dt = L.s(21);
        const V3<T> p = L.v(9) + dt * L.dof(0, 0), e0 = L.v(12) + dt * L.dof(3, 1), e1 = L.v(15) + dt * L.dof(6, 2);
        return 0.5 * k * sqnorm((b0 * e0 + b1 * e1) - p);
    }
};
// :62-85   d_d_p_t   bindings: v1[p,t0,t1,t2]*, x0[p,t0,t1,t2], bary[idx] (3), k[group], dt
struct E_AttachPT
{
    static constexpr const char* name = "EnergyAttachments_d_d_p_t";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double b0 = L.s(24), b1 = L.s(25), b2 = L.s(26), k = L.s(27), dt = L.s(28);
        const V3<T> p = L.v(12) + dt * L.dof(0, 0), t0 = L.v(15) + dt * L.dof(3, 1), t1 = L.v(18) + dt * L.dof(6, 2), t2 = L.v(21) + dt * L.dof(9, 3);
        return 0.5 * k * sqnorm((b0 * t0 + b1 * t1 + b2 * t2) - p);
    }
};
// :87-111  d_d_e_e   bindings: v1[ea0,ea1,eb0,eb1]*, x0[...], bary_0[idx] (2), bary_1[idx] (2), k[group], dt
struct E_AttachEE
{
    static constexpr const char* name = "EnergyAttachments_d_d_e_e";
    using Layout = Strides<3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 1, 1>;
    static constexpr int NB = 4;
    static constexpr int dof_binding[NB] = {0, 1, 2, 3};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double a0 = L.s(24), a1 = L.s(25), b0 = L.s(26), b1 = L.s(27), k = L.s(28), dt = L.s(29);
        const V3<T> ea0 = L.v(12) + dt * L.dof(0, 0), ea1 = L.v(15) + dt * L.dof(3, 1), eb0 = L.v(18) + dt * L.dof(6, 2), eb1 = L.v(21) + dt * L.dof(9, 3);
        return 0.5 * k * sqnorm((b0 * eb0 + b1 * eb1) - (a0 * ea0 + a1 * ea1));
    }
};
// :113-135 rb_d      bindings: k[group], dt, v1_d[p]*, x0_d[p], x_loc[idx], body {v1*, w1*, t0, q0_}[rb]
// local DoF order (DoF sets in registration order): soft.v1 of the point, rigid.v1, rigid.w1
struct E_AttachRBD
{
    static constexpr const char* name = "EnergyAttachments_rb_d";
    using Layout = Strides<1, 1, 3, 3, 3, 3, 3, 3, 4>;
    static constexpr int NB = 3;
    static constexpr int dof_binding[NB] = {2, 5, 6};
    template <class T>
    MS_HD static T energy(const Loader<T>& L)
    {
        const double k = L.s(0), dt = L.s(1);
        const V3<T> x1_d = L.v(5) + dt * L.dof(2, 0);
        const V3<T> x1_rb = rb_point(L, 11, 1, 2, L.v(8), dt);
        return 0.5 * k * sqnorm(x1_d - x1_rb);
    }
};

// element condition: potentials without one are always active
template <class En, class = void>
struct HasCond : std::false_type {};
template <class En>
struct HasCond<En, std::void_t<decltype(En::COND)>> : std::true_type {};
template <class En>
MS_HD bool element_active(const double* in)
{
    if constexpr (HasCond<En>::value) return En::active(in);
    else return true;
}

}  // namespace mistark
