// TEST-ONLY host build of the element energies (stark_amd/csrc/energies.hpp compiled with g++): lets the CPU test suite
// check the exact device math (same header, hyper-dual seeding included) against the oracle and the golden fixtures
// without a GPU. Not part of the product library.
#include <cstring>
#include "../../stark_amd/csrc/registry.hpp"
#include "../../stark_amd/csrc/tet_closed.hpp"

using namespace mistark;

template <class En>
static void eval(const double* in, int n_elem, double* E, double* g, double* H)
{
    constexpr int NIN = En::Layout::NIN;
    constexpr int n = 3 * En::NB;
    for (int e = 0; e < n_elem; e++) {
        const double* ie = in + (size_t)e * NIN;
        Loader<double> Ld{ie};
        E[e] = En::energy(Ld);
        for (int i = 0; i < n; i++) {
            for (int j = i; j < n; j++) {
                Loader<HDual> L{ie, i, j};
                const HDual r = En::energy(L);
                H[((size_t)e * n + i) * n + j] = r.ab;
                H[((size_t)e * n + j) * n + i] = r.ab;
                if (i == j) g[(size_t)e * n + i] = r.a;
            }
        }
    }
}

extern "C" int host_elem_info(const char* name, int* nb, int* nin, int* nbind, int* strides)
{
#define X(En)                                                            \
    if (std::strcmp(name, En::name) == 0) {                              \
        *nb = En::NB; *nin = En::Layout::NIN; *nbind = En::Layout::NBIND; \
        En::Layout::strides(strides);                                    \
        return 0;                                                        \
    }
    MISTARK_FOR_EACH_ENERGY(X)
#undef X
    return -1;
}

extern "C" int host_elem_eval(const char* name, const double* in, int n_elem, double* E, double* g, double* H)
{
#define X(En) \
    if (std::strcmp(name, En::name) == 0) { eval<En>(in, n_elem, E, g, H); return 0; }
    MISTARK_FOR_EACH_ENERGY(X)
#undef X
    return -1;
}

// closed-form tet kernels (stark_amd/csrc/tet_closed.hpp); H returned row-major 12x12 like host_elem_eval
extern "C" int host_tet_closed_eval(int full, const double* in, int n_elem, double* E, double* g, double* H)
{
    const int NIN = full ? E_TetStrain::Layout::NIN : E_TetStrainEO::Layout::NIN;
    for (int e = 0; e < n_elem; e++) {
        double Hb[144];
        if (full) tet_closed_eval<true>(in + (size_t)e * NIN, E[e], g + 12 * (size_t)e, Hb, 9, true);
        else tet_closed_eval<false>(in + (size_t)e * NIN, E[e], g + 12 * (size_t)e, Hb, 9, true);
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++)
                for (int i = 0; i < 3; i++)
                    for (int k = 0; k < 3; k++) H[((size_t)e * 12 + 3 * a + i) * 12 + 3 * b + k] = Hb[(a * 4 + b) * 9 + 3 * i + k];
    }
    return 0;
}

// ---- narrow-phase geometry of the contact detector (stark_amd/csrc/contact_geom.hpp), batch wrappers ------------------------
#include "../../stark_amd/csrc/contact_geom.hpp"
static D3 ld(const double* x) { return d3(x[0], x[1], x[2]); }
extern "C" void host_geom_point_triangle(const double* in, int n, int* type, double* d2)
{
    for (int i = 0; i < n; i++) d2[i] = point_triangle_sq_distance(type[i], ld(in + 12 * i), ld(in + 12 * i + 3), ld(in + 12 * i + 6), ld(in + 12 * i + 9));
}
extern "C" void host_geom_edge_edge(const double* in, int n, int* type, double* d2)
{
    for (int i = 0; i < n; i++) d2[i] = edge_edge_sq_distance(type[i], ld(in + 12 * i), ld(in + 12 * i + 3), ld(in + 12 * i + 6), ld(in + 12 * i + 9));
}
extern "C" void host_geom_edge_triangle(const double* in, int n, int* hit)
{
    for (int i = 0; i < n; i++) hit[i] = edge_intersects_triangle(ld(in + 15 * i), ld(in + 15 * i + 3), ld(in + 15 * i + 6), ld(in + 15 * i + 9), ld(in + 15 * i + 12)) ? 1 : 0;
}
// out: pt [bary3 | T6], pe [bary2 | T6], pp [T6] from the point-triangle inputs; ee [bary2 | T6] from the edge-edge inputs
extern "C" void host_geom_friction(const double* pt_in, const double* ee_in, int n, double* pt, double* pe, double* pp, double* ee)
{
    for (int i = 0; i < n; i++) {
        const D3 p = ld(pt_in + 12 * i), a = ld(pt_in + 12 * i + 3), b = ld(pt_in + 12 * i + 6), c = ld(pt_in + 12 * i + 9);
        bary_point_triangle(p, a, b, c, pt + 9 * i);
        basis_triangle(a, b, c, pt + 9 * i + 3);
        bary_point_edge(p, a, b, pe + 8 * i);
        basis_point_edge(p, a, b, pe + 8 * i + 2);
        basis_point_point(p, a, pp + 6 * i);
        const D3 A = ld(ee_in + 12 * i), B = ld(ee_in + 12 * i + 3), P = ld(ee_in + 12 * i + 6), Q = ld(ee_in + 12 * i + 9);
        bary_edge_edge(A, B, P, Q, ee + 8 * i);
        basis_edge_edge(A, B, P, Q, ee + 8 * i + 2);
    }
}
