// tet_closed.hpp — hand-derived closed-form energy / gradient / Hessian of the volumetric strain potentials
//   EnergyTetStrain_Elasticity_Only  (stark/src/models/deformables/volume/EnergyTetStrain.cpp:80-123)
//   EnergyTetStrain                  (stark/src/models/deformables/volume/EnergyTetStrain.cpp:12-78)
// replacing the reference's 1 896 / 12 736-operation generated kernels by ~2 k flops per tet.
//
// Notation: x_a = x0_a + dt v_a (a = 0..3), F = sum_a x_a w_a^T with the constant shape-function gradients
// w_a (rows of DX^-1, w_0 = -(w_1+w_2+w_3)), V = det(DX)/6, psi(F) = psi_el(F) + phi(E), E = (F^T F - I)/2.
//   psi_el = mu'/2 (Ic-3) + lambda'/2 (J-alpha)^2 - mu'/2 log(Ic+1)                       (stable Neo-Hookean, Smith et al.)
//   phi    = c/(2 dt^2) |E - E0|^2 + [dl>0] k/3 dl^3,  dl = tr E/3 + sqrt(2/3) |dev E| - limit   (damping + strain limiting)
// First Piola stress  P = c1 F + lambda'(J-alpha) cof F + F T,      T = dphi/dE (symmetric)
// Hessian block (a,b), rows i / cols k, before the V dt^2 factor:
//   [c1 w_a.w_b + w_a.T w_b] I + c2 (F w_a)(F w_b)^T + lambda' (C w_a)(C w_b)^T + lambda'(J-alpha) eps_ikm (F (w_a x w_b))_m
//   + kappa1/2 [ (F F^T) (w_a.w_b) + (F w_b)(F w_a)^T ] + kappa2 (F G w_a)(F G w_b)^T - kappa3/3 (F w_a)(F w_b)^T - kappa4 (F D w_a)(F D w_b)^T
// with c1 = mu'(1 - 1/(Ic+1)), c2 = 2 mu'/(Ic+1)^2, kappa1 = c/dt^2 + kappa3, kappa2 = 2 k dl, kappa3 = k dl^2 sqrt(2/3)/|D|,
// kappa4 = kappa3/|D|^2, G = I/3 + sqrt(2/3) D/|D|, D = dev E.
#pragma once
#include <cstddef>

#include "hdual.hpp"

namespace mistark {

// in[]: gathered inputs in the binding order of the reference (same layout as E_TetStrain / E_TetStrainEO in energies.hpp):
//   v1[4] (0..11), x0[4] (12..23), X[4] (24..35), then EO: scale, e, nu, dt | FULL: scale, e, nu, strain_limit, strain_limit_stiffness, damping, dt
// out: E (energy), g[12] (dE/dv), 3x3 blocks (a,b) row-major at H + (4a+b)*hstride (only if want_h)
template <bool FULL, class Sink>
MS_HD void tet_closed_eval_to(const double* in, double& E_out, double* g, Sink& sink, bool want_h)
{
    const double scale = in[36], e = in[37], nu = in[38];
    const double strain_limit = FULL ? in[39] : 0.0, sl_k = FULL ? in[40] : 0.0, damping = FULL ? in[41] : 0.0;
    const double dt = FULL ? in[42] : in[39];

    // rest shape
    double DX[3][3];
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) DX[i][k] = scale * (in[24 + 3 * (k + 1) + i] - in[24 + i]);
    const double c00 = DX[1][1] * DX[2][2] - DX[1][2] * DX[2][1];
    const double c01 = DX[1][2] * DX[2][0] - DX[1][0] * DX[2][2];
    const double c02 = DX[1][0] * DX[2][1] - DX[1][1] * DX[2][0];
    const double detDX = DX[0][0] * c00 + DX[0][1] * c01 + DX[0][2] * c02;
    const double idet = 1.0 / detDX;
    double w[4][3];  // w[a][j], a = 1..3: row a-1 of DX^-1
    w[1][0] = c00 * idet;
    w[1][1] = (DX[0][2] * DX[2][1] - DX[0][1] * DX[2][2]) * idet;
    w[1][2] = (DX[0][1] * DX[1][2] - DX[0][2] * DX[1][1]) * idet;
    w[2][0] = c01 * idet;
    w[2][1] = (DX[0][0] * DX[2][2] - DX[0][2] * DX[2][0]) * idet;
    w[2][2] = (DX[0][2] * DX[1][0] - DX[0][0] * DX[1][2]) * idet;
    w[3][0] = c02 * idet;
    w[3][1] = (DX[0][1] * DX[2][0] - DX[0][0] * DX[2][1]) * idet;
    w[3][2] = (DX[0][0] * DX[1][1] - DX[0][1] * DX[1][0]) * idet;
    for (int j = 0; j < 3; j++) w[0][j] = -(w[1][j] + w[2][j] + w[3][j]);
    const double vol = detDX / 6.0;

    // deformation gradient(s)
    double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, F0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 3; i++) {
            const double x0 = in[12 + 3 * a + i];
            const double x1 = x0 + dt * in[3 * a + i];
            for (int j = 0; j < 3; j++) {
                F[i][j] += x1 * w[a][j];
                if (FULL) F0[i][j] += x0 * w[a][j];
            }
        }
    // cofactor C = dJ/dF: columns f1 x f2, f2 x f0, f0 x f1
    double C[3][3];
    for (int j = 0; j < 3; j++) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        C[0][j] = F[1][j1] * F[2][j2] - F[2][j1] * F[1][j2];
        C[1][j] = F[2][j1] * F[0][j2] - F[0][j1] * F[2][j2];
        C[2][j] = F[0][j1] * F[1][j2] - F[1][j1] * F[0][j2];
    }
    const double J = F[0][0] * C[0][0] + F[1][0] * C[1][0] + F[2][0] * C[2][0];
    double Ic = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ic += F[i][j] * F[i][j];

    const double mu = e / (2.0 * (1.0 + nu));
    const double lambda = (e * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const double mu_ = 4.0 / 3.0 * mu;
    const double lambda_ = lambda + 5.0 / 6.0 * mu;
    const double alpha = 1.0 + mu_ / lambda_ - mu_ / (4.0 * lambda_);
    const double Jm = J - alpha;
    double psi = 0.5 * mu_ * (Ic - 3.0) + 0.5 * lambda_ * Jm * Jm - 0.5 * mu_ * ::log(Ic + 1.0);
    const double ir = 1.0 / (Ic + 1.0);
    const double c1 = mu_ * (1.0 - ir);
    const double c2 = 2.0 * mu_ * ir * ir;
    const double c3 = lambda_ * Jm;

    // phi(E): damping + strain limiting
    double T[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // dphi/dE
    double G[3][3], D[3][3];
    double kap1 = 0.0, kap2 = 0.0, kap3 = 0.0, kap4 = 0.0, gam_n = 0.0;
    bool limiting = false;
    if (FULL) {
        double E1[3][3], E0[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s1 = 0.0, s0 = 0.0;
                for (int m = 0; m < 3; m++) {
                    s1 += F[m][i] * F[m][j];
                    s0 += F0[m][i] * F0[m][j];
                }
                E1[i][j] = 0.5 * (s1 - (i == j ? 1.0 : 0.0));
                E0[i][j] = 0.5 * (s0 - (i == j ? 1.0 : 0.0));
            }
        const double cd = damping / (dt * dt);
        double ss = 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double s = E1[i][j] - E0[i][j];
                ss += s * s;
                T[i][j] = cd * s;
            }
        psi += 0.5 * cd * ss;
        kap1 = cd;
        const double trE = E1[0][0] + E1[1][1] + E1[2][2];
        double n2 = 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                D[i][j] = E1[i][j] - (i == j ? trE / 3.0 : 0.0);
                n2 += D[i][j] * D[i][j];
            }
        const double gam = 0.81649658092772603;  // sqrt(2/3)
        const double n = ::sqrt(n2);
        const double dl = trE / 3.0 + gam * n - strain_limit;
        if (dl > 0.0) {
            limiting = true;
            psi += sl_k * dl * dl * dl / 3.0;
            const double in_ = 1.0 / n;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    G[i][j] = (i == j ? 1.0 / 3.0 : 0.0) + gam * D[i][j] * in_;
                    T[i][j] += sl_k * dl * dl * G[i][j];
                }
            kap2 = 2.0 * sl_k * dl;
            kap3 = sl_k * dl * dl * gam * in_;
            kap4 = kap3 * in_ * in_;
            gam_n = gam * in_;
            kap1 += kap3;
        }
    }
    E_out = vol * psi;

    // per-node vectors
    double fa[4][3], ca[4][3], ta[4][3], da[4][3];
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 3; i++) {
            fa[a][i] = F[i][0] * w[a][0] + F[i][1] * w[a][1] + F[i][2] * w[a][2];
            ca[a][i] = C[i][0] * w[a][0] + C[i][1] * w[a][1] + C[i][2] * w[a][2];
            if (FULL) ta[a][i] = T[i][0] * w[a][0] + T[i][1] * w[a][1] + T[i][2] * w[a][2];
        }
    // strain limiting: d_a = F D w_a; g_a = F G w_a = f_a / 3 + (gam / n) d_a is formed per block from f_a and d_a (G = I / 3 + gam D / n)
    if (FULL && limiting) {
        for (int a = 0; a < 4; a++) {
            double dw[3];
            for (int i = 0; i < 3; i++) dw[i] = D[i][0] * w[a][0] + D[i][1] * w[a][1] + D[i][2] * w[a][2];
            for (int i = 0; i < 3; i++) da[a][i] = F[i][0] * dw[0] + F[i][1] * dw[1] + F[i][2] * dw[2];
        }
    }
    // gradient: g_a = V dt (P w_a),  P w_a = c1 F w_a + c3 C w_a + F (T w_a)
    const double sg = vol * dt;
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 3; i++) {
            double v = c1 * fa[a][i] + c3 * ca[a][i];
            if (FULL) v += F[i][0] * ta[a][0] + F[i][1] * ta[a][1] + F[i][2] * ta[a][2];
            g[3 * a + i] = sg * v;
        }
    if (!want_h) return;
    sink.energy_and_gradient(E_out, g);  // (a sink that stores them here frees their registers for the block loop)

    double FFt[3][3];
    if (FULL)
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) FFt[i][k] = F[i][0] * F[k][0] + F[i][1] * F[k][1] + F[i][2] * F[k][2];
    const double sh = vol * dt * dt, third = 1.0 / 3.0;
    for (int a = 0; a < 4; a++)
        for (int b = a; b < 4; b++) {
            const double wab = w[a][0] * w[b][0] + w[a][1] * w[b][1] + w[a][2] * w[b][2];
            double diag = c1 * wab;
            if (FULL) diag += w[a][0] * ta[b][0] + w[a][1] * ta[b][1] + w[a][2] * ta[b][2];
            // q = F (w_a x w_b)
            const double cx = w[a][1] * w[b][2] - w[a][2] * w[b][1], cy = w[a][2] * w[b][0] - w[a][0] * w[b][2], cz = w[a][0] * w[b][1] - w[a][1] * w[b][0];
            double q[3];
            for (int m = 0; m < 3; m++) q[m] = c3 * (F[m][0] * cx + F[m][1] * cy + F[m][2] * cz);
            double M[3][3];
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++) {
                    double v = c2 * fa[a][i] * fa[b][k] + lambda_ * ca[a][i] * ca[b][k];
                    if (FULL) {
                        v += 0.5 * kap1 * (FFt[i][k] * wab + fa[b][i] * fa[a][k]);
                        if (limiting) v += kap2 * (third * fa[a][i] + gam_n * da[a][i]) * (third * fa[b][k] + gam_n * da[b][k]) - (kap3 / 3.0) * fa[a][i] * fa[b][k] - kap4 * da[a][i] * da[b][k];
                    }
                    M[i][k] = v;
                }
            M[0][0] += diag;
            M[1][1] += diag;
            M[2][2] += diag;
            // eps_ikm q_m
            M[0][1] += q[2];
            M[1][0] -= q[2];
            M[1][2] += q[0];
            M[2][1] -= q[0];
            M[2][0] += q[1];
            M[0][2] -= q[1];
            double blk[9];
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++) blk[3 * i + k] = sh * M[i][k];
            sink.put(a, b, blk);  // block (a, b), a <= b; the sink also places the transposed block (b, a)
        }
}
// blocks straight to memory: (a, b) row-major at H + (4a+b)*hstride
struct TetBlockMemSink
{
    double* H;
    size_t hstride;
    MS_HD void energy_and_gradient(double, const double*) {}
    MS_HD void put(int a, int b, const double* blk)
    {
        double* Hab = H + (size_t)(a * 4 + b) * hstride;
        double* Hba = H + (size_t)(b * 4 + a) * hstride;
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) {
                Hab[3 * i + k] = blk[3 * i + k];
                Hba[3 * k + i] = blk[3 * i + k];
            }
    }
};
template <bool FULL>
MS_HD void tet_closed_eval(const double* in, double& E_out, double* g, double* H, size_t hstride, bool want_h)
{
    TetBlockMemSink sink{H, hstride};
    tet_closed_eval_to<FULL>(in, E_out, g, sink, want_h);
}

}  // namespace mistark
