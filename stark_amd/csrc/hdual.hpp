// hdual.hpp — scalar types the element energies are written against.
//
// The reference differentiates every energy symbolically and JIT-compiles one straight-line kernel per potential
// (symx/src/solver/second_order/SecondOrderCompiledPotential.cpp:64-85). On CDNA4 we instead evaluate the *energy
// expression itself* on hyper-dual numbers, one (i,j) derivative pair per lane: a lane that seeds eps1 on DoF i and
// eps2 on DoF j obtains E, dE/du_i, dE/du_j and d2E/du_i du_j in 4 doubles per live scalar (register-light, no
// spilling of n x n Hessians), and the n(n+1)/2 pairs of an element are spread over the lanes of a wavefront.
// The hot volumetric potentials additionally have hand-derived closed-form kernels (tet_closed.hpp).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif

#if defined(__HIPCC__)
#define MS_HD __host__ __device__ __forceinline__
#else
#define MS_HD inline
#endif

namespace mistark {

// f(x + a e1 + b e2) = v + a e1 + b e2 + ab e1e2,  e1^2 = e2^2 = 0
struct HDual
{
    double v, a, b, ab;
    MS_HD HDual() : v(0), a(0), b(0), ab(0) {}
    MS_HD HDual(double v_) : v(v_), a(0), b(0), ab(0) {}
    MS_HD HDual(double v_, double a_, double b_, double ab_) : v(v_), a(a_), b(b_), ab(ab_) {}
};

MS_HD HDual operator+(const HDual& x, const HDual& y) { return HDual(x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab); }
MS_HD HDual operator-(const HDual& x, const HDual& y) { return HDual(x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab); }
MS_HD HDual operator-(const HDual& x) { return HDual(-x.v, -x.a, -x.b, -x.ab); }
MS_HD HDual operator*(const HDual& x, const HDual& y)
{
    return HDual(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab);
}
MS_HD HDual operator+(const HDual& x, double y) { return HDual(x.v + y, x.a, x.b, x.ab); }
MS_HD HDual operator+(double y, const HDual& x) { return HDual(x.v + y, x.a, x.b, x.ab); }
MS_HD HDual operator-(const HDual& x, double y) { return HDual(x.v - y, x.a, x.b, x.ab); }
MS_HD HDual operator-(double y, const HDual& x) { return HDual(y - x.v, -x.a, -x.b, -x.ab); }
MS_HD HDual operator*(const HDual& x, double y) { return HDual(x.v * y, x.a * y, x.b * y, x.ab * y); }
MS_HD HDual operator*(double y, const HDual& x) { return HDual(x.v * y, x.a * y, x.b * y, x.ab * y); }

// y = f(x) given f, f', f'' at x.v
MS_HD HDual chain(const HDual& x, double f, double df, double ddf)
{
    return HDual(f, df * x.a, df * x.b, df * x.ab + ddf * x.a * x.b);
}
MS_HD HDual inv(const HDual& x)
{
    const double r = 1.0 / x.v;
    return chain(x, r, -r * r, 2.0 * r * r * r);
}
// double overloads (unqualified calls inside this namespace would otherwise convert to HDual)
MS_HD double sqrt(double x) { return ::sqrt(x); }
MS_HD double log(double x) { return ::log(x); }
MS_HD double acos(double x) { return ::acos(x); }
MS_HD double atan(double x) { return ::atan(x); }
MS_HD double cos(double x) { return ::cos(x); }
MS_HD double sin(double x) { return ::sin(x); }
MS_HD HDual operator/(const HDual& x, const HDual& y) { return x * inv(y); }
MS_HD HDual operator/(const HDual& x, double y) { return x * (1.0 / y); }
MS_HD HDual operator/(double x, const HDual& y) { return x * inv(y); }
MS_HD HDual sqrt(const HDual& x)
{
    const double s = ::sqrt(x.v);
    return chain(x, s, 0.5 / s, -0.25 / (s * x.v));
}
MS_HD HDual log(const HDual& x)
{
    const double r = 1.0 / x.v;
    return chain(x, ::log(x.v), r, -r * r);
}
MS_HD HDual acos(const HDual& x)
{
    const double s = 1.0 / ::sqrt(1.0 - x.v * x.v);
    return chain(x, ::acos(x.v), -s, -x.v * s * s * s);
}
MS_HD HDual atan(const HDual& x)
{
    const double d = 1.0 / (1.0 + x.v * x.v);
    return chain(x, ::atan(x.v), d, -2.0 * x.v * d * d);
}
MS_HD HDual cos(const HDual& x) { return chain(x, ::cos(x.v), -::sin(x.v), -::cos(x.v)); }
MS_HD HDual sin(const HDual& x) { return chain(x, ::sin(x.v), ::cos(x.v), -::sin(x.v)); }
MS_HD HDual pow2(const HDual& x) { return x * x; }
MS_HD HDual pow3(const HDual& x) { return x * x * x; }
MS_HD double pow2(double x) { return x * x; }
MS_HD double pow3(double x) { return x * x * x; }
MS_HD double val(const HDual& x) { return x.v; }
MS_HD double val(double x) { return x; }
MS_HD double inv(double x) { return 1.0 / x; }

// branch(cond, a, b) of the reference (value and derivatives of the selected side only)
MS_HD HDual select(bool c, const HDual& x, const HDual& y) { return c ? x : y; }
MS_HD double select(bool c, double x, double y) { return c ? x : y; }

// ---- tiny fixed-size vectors / matrices over any scalar ---------------------------------------------------------------
template <class T>
struct V3
{
    T x, y, z;
    MS_HD V3() {}
    MS_HD V3(const T& x_, const T& y_, const T& z_) : x(x_), y(y_), z(z_) {}
    MS_HD T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    MS_HD const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class A, class B>
struct promote { using type = HDual; };
template <>
struct promote<double, double> { using type = double; };
template <class A, class B>
using promote_t = typename promote<A, B>::type;

template <class A, class B>
MS_HD V3<promote_t<A, B>> operator+(const V3<A>& a, const V3<B>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class A, class B>
MS_HD V3<promote_t<A, B>> operator-(const V3<A>& a, const V3<B>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class A>
MS_HD V3<A> operator-(const V3<A>& a) { return {-a.x, -a.y, -a.z}; }
template <class B>
MS_HD V3<B> operator*(double s, const V3<B>& b) { return {s * b.x, s * b.y, s * b.z}; }
template <class B>
MS_HD V3<HDual> operator*(const HDual& s, const V3<B>& b) { return {s * b.x, s * b.y, s * b.z}; }
template <class A, class B>
MS_HD promote_t<A, B> dot(const V3<A>& a, const V3<B>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class A, class B>
MS_HD V3<promote_t<A, B>> cross(const V3<A>& a, const V3<B>& b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class A>
MS_HD A sqnorm(const V3<A>& a) { return dot(a, a); }
template <class A>
MS_HD A norm(const V3<A>& a) { return sqrt(dot(a, a)); }
template <class A>
MS_HD V3<A> normalized(const V3<A>& a)
{
    const A r = inv(norm(a));
    return {a.x * r, a.y * r, a.z * r};
}

template <class T>
struct M3
{
    T m[3][3];
    MS_HD T& operator()(int i, int j) { return m[i][j]; }
    MS_HD const T& operator()(int i, int j) const { return m[i][j]; }
};
template <class T>
MS_HD M3<T> from_cols(const V3<T>& c0, const V3<T>& c1, const V3<T>& c2)
{
    M3<T> r;
    for (int i = 0; i < 3; i++) { r.m[i][0] = c0[i]; r.m[i][1] = c1[i]; r.m[i][2] = c2[i]; }
    return r;
}
template <class T>
MS_HD T det(const M3<T>& A)
{
    return A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) - A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0])
           + A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
}
template <class T>
MS_HD M3<T> inverse(const M3<T>& A)
{
    const T r = inv(det(A));
    M3<T> c;
    c.m[0][0] = (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) * r;
    c.m[0][1] = (A.m[0][2] * A.m[2][1] - A.m[0][1] * A.m[2][2]) * r;
    c.m[0][2] = (A.m[0][1] * A.m[1][2] - A.m[0][2] * A.m[1][1]) * r;
    c.m[1][0] = (A.m[1][2] * A.m[2][0] - A.m[1][0] * A.m[2][2]) * r;
    c.m[1][1] = (A.m[0][0] * A.m[2][2] - A.m[0][2] * A.m[2][0]) * r;
    c.m[1][2] = (A.m[0][2] * A.m[1][0] - A.m[0][0] * A.m[1][2]) * r;
    c.m[2][0] = (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]) * r;
    c.m[2][1] = (A.m[0][1] * A.m[2][0] - A.m[0][0] * A.m[2][1]) * r;
    c.m[2][2] = (A.m[0][0] * A.m[1][1] - A.m[0][1] * A.m[1][0]) * r;
    return c;
}
template <class A, class B>
MS_HD M3<promote_t<A, B>> operator*(const M3<A>& a, const M3<B>& b)
{
    M3<promote_t<A, B>> r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
template <class A, class B>
MS_HD V3<promote_t<A, B>> operator*(const M3<A>& a, const V3<B>& b)
{
    return {a.m[0][0] * b.x + a.m[0][1] * b.y + a.m[0][2] * b.z, a.m[1][0] * b.x + a.m[1][1] * b.y + a.m[1][2] * b.z,
            a.m[2][0] * b.x + a.m[2][1] * b.y + a.m[2][2] * b.z};
}
template <class T>
MS_HD M3<T> transpose(const M3<T>& a)
{
    M3<T> r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
template <class T>
MS_HD T frob_sq(const M3<T>& a)
{
    T s = a.m[0][0] * a.m[0][0];
    for (int k = 1; k < 9; k++) s = s + a.m[k / 3][k % 3] * a.m[k / 3][k % 3];
    return s;
}
template <class T>
MS_HD T trace(const M3<T>& a) { return a.m[0][0] + a.m[1][1] + a.m[2][2]; }

}  // namespace mistark
