// tri_closed.hpp — the membrane potentials through their invariants instead of one hyper-dual energy evaluation per Hessian entry:
//   EnergyTriangleStrain                  (stark/src/models/deformables/surface/EnergyTriangleStrain.cpp:13-80)
//   EnergyTriangleStrain_Elasticity_Only  (:82-129)
// E(v) = w psi(c(x)) + w (inflation / 3) n0 . (x_0 + x_1 + x_2),  x_a = x0_a + dt v_a,  w = thickness * rest area,
// c = (C00, C01, C11) the entries of C = F^T F with F = [f0 | f1] = sum_a x_a b_a^T (3 x 2; b_a = in-plane shape-function gradients of the
// rest triangle, b_0 = -(b_1 + b_2)). psi only sees x through the three numbers c, and c is QUADRATIC in x:
//   dc/dx_a      = (2 b_a0 f0,  b_a0 f1 + b_a1 f0,  2 b_a1 f1)                       (three 3-vectors per node)
//   d2c/dx_a dx_b = (2 b_a0 b_b0,  b_a0 b_b1 + b_a1 b_b0,  2 b_a1 b_b1) I3            (constants)
//   dE/dx_a  = w sum_k psi_k dc_k/dx_a + w (inflation / 3) n0
//   d2E/dx_a dx_b = w [ sum_kl psi_kl (dc_k/dx_a)(dc_l/dx_b)^T + sum_k psi_k d2c_k/dx_a dx_b ]
// The first and second derivatives of psi (3 + 6 numbers) come from SIX hyper-dual evaluations of a function of three scalars (log,
// square root, the damping and the two strain-limiting terms) instead of 45 evaluations of the whole energy with its cross products and
// rest-shape algebra. psi itself is the reference's expression from C on; J = area / rest area enters as sqrt(det C) (the same number:
// |d1 x d2|^2 = det(D^T D)).
#pragma once
#include "hdual.hpp"

namespace mistark {

struct TriParams
{
    double mu, lambda, damping, idt, P00, P01, P11, strain_limit, sl_k;
};
template <bool FULL, class T>
MS_HD T tri_density(const T& C00, const T& C01, const T& C11, const TriParams& p)
{
    const T J = sqrt(C00 * C11 - C01 * C01);
    const T Ic = C00 + C11;
    const T logJ = log(J);
    T density = 0.5 * p.mu * (Ic - 2.0) - p.mu * logJ + 0.5 * p.lambda * pow2(logJ);
    if (FULL) {
        const T E00 = 0.5 * (C00 - 1.0), E01 = 0.5 * C01, E11 = 0.5 * (C11 - 1.0);
        density = density + 0.5 * p.damping * (pow2((E00 - p.P00) * p.idt) + 2.0 * pow2((E01 - p.P01) * p.idt) + pow2((E11 - p.P11) * p.idt));
        const T disc = 4.0 * pow2(E01) + pow2(E00 - E11);
        const double sq = ::sqrt(val(disc));
        const double s0 = 0.5 * (val(E00) + val(E11) + sq), s1 = 0.5 * (val(E00) + val(E11) - sq);
        if (s0 - p.strain_limit > 0.0) density = density + (p.sl_k / 3.0) * pow3(0.5 * (E00 + E11 + sqrt(disc)) - p.strain_limit);
        if (s1 - p.strain_limit > 0.0) density = density + (p.sl_k / 3.0) * pow3(0.5 * (E00 + E11 - sqrt(disc)) - p.strain_limit);
    }
    return density;
}

// in[]: the gathered inputs in binding order (E_TriangleStrain / E_TriangleStrainEO in energies.hpp): v1[3] (0..8), x0[3] (9..17), X[3] (18..26),
// scale, thickness, e, nu, then FULL: strain_damping, strain_limit, strain_limit_stiffness, inflation, dt | EO: inflation, dt.
// out: E, g[9] = dE/dv, H[a][b][9] = 3x3 blocks d2E/dv_a dv_b (row-major), all nine blocks (only if want_h)
template <bool FULL>
MS_HD void tri_closed_eval(const double* in, double& E_out, double* g, double (*H)[3][9], bool want_h)
{
    const int p = 27;
    const double scale = in[p], thickness = in[p + 1], e = in[p + 2], nu = in[p + 3];
    const double damping = FULL ? in[p + 4] : 0.0, strain_limit = FULL ? in[p + 5] : 0.0, sl_k = FULL ? in[p + 6] : 0.0;
    const double inflation = in[FULL ? p + 7 : p + 4], dt = in[FULL ? p + 8 : p + 5];
    V3<double> x1[3], x0[3], Xs[3];
    for (int i = 0; i < 3; i++) {
        x0[i] = V3<double>(in[9 + 3 * i], in[10 + 3 * i], in[11 + 3 * i]);
        x1[i] = V3<double>(x0[i].x + dt * in[3 * i], x0[i].y + dt * in[3 * i + 1], x0[i].z + dt * in[3 * i + 2]);
        Xs[i] = V3<double>(scale * in[18 + 3 * i], scale * in[19 + 3 * i], scale * in[20 + 3 * i]);
    }
    const double rest_area = 0.5 * norm(cross(Xs[0] - Xs[2], Xs[1] - Xs[2]));
    // rest configuration projected into its own plane -> 2x2 Jacobian and its inverse (deformable_tools.cpp:7-21)
    const V3<double> u = normalized(Xs[1] - Xs[0]);
    const V3<double> n = cross(u, Xs[2] - Xs[0]);
    const V3<double> v = normalized(cross(u, n));
    const double a00 = dot(u, Xs[1]) - dot(u, Xs[0]), a01 = dot(u, Xs[2]) - dot(u, Xs[0]);
    const double a10 = dot(v, Xs[1]) - dot(v, Xs[0]), a11 = dot(v, Xs[2]) - dot(v, Xs[0]);
    const double idet = 1.0 / (a00 * a11 - a01 * a10);
    const double i00 = a11 * idet, i01 = -a01 * idet, i10 = -a10 * idet, i11 = a00 * idet;
    // F = [f0 | f1] = sum_a x_a (b_a0, b_a1): d1 = x_1 - x_0, d2 = x_2 - x_0, f0 = i00 d1 + i10 d2, f1 = i01 d1 + i11 d2
    const double b[3][2] = {{-(i00 + i10), -(i01 + i11)}, {i00, i01}, {i10, i11}};
    const V3<double> d1 = x1[1] - x1[0], d2 = x1[2] - x1[0];
    const V3<double> f0 = i00 * d1 + i10 * d2, f1 = i01 * d1 + i11 * d2;
    const double C00 = dot(f0, f0), C01 = dot(f0, f1), C11 = dot(f1, f1);
    TriParams P{};
    P.mu = e / (2.0 * (1.0 + nu));
    P.lambda = (e * nu) / ((1.0 + nu) * (1.0 - nu));  // 2D
    P.damping = damping;
    P.strain_limit = strain_limit;
    P.sl_k = sl_k;
    if (FULL) {
        const V3<double> e1 = x0[1] - x0[0], e2 = x0[2] - x0[0];
        const V3<double> g0 = i00 * e1 + i10 * e2, g1 = i01 * e1 + i11 * e2;
        P.P00 = 0.5 * (dot(g0, g0) - 1.0);
        P.P01 = 0.5 * dot(g0, g1);
        P.P11 = 0.5 * (dot(g1, g1) - 1.0);
        P.idt = 1.0 / dt;
    }
    // psi, its gradient and Hessian w.r.t. (C00, C01, C11): hyper-dual pairs (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    double psi = 0.0, dpsi[3] = {0, 0, 0}, hpsi[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            if (!want_h && i != j) continue;
            const HDual c0(C00, i == 0 ? 1.0 : 0.0, j == 0 ? 1.0 : 0.0, 0.0), c1(C01, i == 1 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, 0.0), c2(C11, i == 2 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0, 0.0);
            const HDual r = tri_density<FULL>(c0, c1, c2, P);
            psi = r.v;
            if (i == j) dpsi[i] = r.a;
            hpsi[i][j] = hpsi[j][i] = r.ab;
        }
    const double w = thickness * rest_area;
    const V3<double> n0 = -normalized(cross(x0[1] - x0[0], x0[2] - x0[0]));
    E_out = w * (psi + (inflation / 3.0) * dot(n0, x1[0] + x1[1] + x1[2]));
    // dc_k/dx_a
    V3<double> dc[3][3];  // [node][k]
    for (int a = 0; a < 3; a++) {
        dc[a][0] = (2.0 * b[a][0]) * f0;
        dc[a][1] = b[a][0] * f1 + b[a][1] * f0;
        dc[a][2] = (2.0 * b[a][1]) * f1;
    }
    const double sg = w * dt;
    for (int a = 0; a < 3; a++) {
        const V3<double> ga = dpsi[0] * dc[a][0] + dpsi[1] * dc[a][1] + dpsi[2] * dc[a][2] + (inflation / 3.0) * n0;
        g[3 * a] = sg * ga.x;
        g[3 * a + 1] = sg * ga.y;
        g[3 * a + 2] = sg * ga.z;
    }
    if (!want_h) return;
    const double sh = w * dt * dt;
    for (int a = 0; a < 3; a++) {
        // t_l = sum_k psi_kl dc_k/dx_a
        V3<double> t[3];
        for (int l = 0; l < 3; l++) t[l] = hpsi[0][l] * dc[a][0] + hpsi[1][l] * dc[a][1] + hpsi[2][l] * dc[a][2];
        for (int bb = 0; bb < 3; bb++) {
            const double diag = dpsi[0] * 2.0 * b[a][0] * b[bb][0] + dpsi[1] * (b[a][0] * b[bb][1] + b[a][1] * b[bb][0]) + dpsi[2] * 2.0 * b[a][1] * b[bb][1];
            double* blk = H[a][bb];
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++) {
                    double s = t[0][i] * dc[bb][0][k] + t[1][i] * dc[bb][1][k] + t[2][i] * dc[bb][2][k];
                    if (i == k) s += diag;
                    blk[3 * i + k] = sh * s;
                }
        }
    }
}

}  // namespace mistark
