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
