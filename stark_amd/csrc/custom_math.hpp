// custom_math.hpp — the operations of a SymX op sequence (symx/src/symbol/Expr.h:12-43; scalar semantics as the reference's emitter writes them,
// symx/src/compile/Compilation.cpp:381-469) on hyper-dual numbers. ONE definition for both ways a user-defined potential runs on the device:
// the interpreter of custom.hip reads the sequence from memory and calls these, the kernels custom.hip EMITS through hipRTC are straight-line
// calls of the same functions — the two can only differ in how the compiler schedules them. Needs hdual.hpp; no other includes (hipRTC).
#pragma once
#include "hdual.hpp"

namespace mistark {

__device__ __forceinline__ HDual cop_pown(const HDual& x, int n)
{
    if (n == 0) return HDual(1.0);
    const double p2 = ::pow(x.v, (double)(n - 2)), p1 = p2 * x.v;  // x^(n-2), x^(n-1)
    if (n == 1) return x;
    if (n == 2) return x * x;
    return chain(x, p1 * x.v, n * p1, (double)n * (n - 1) * p2);
}
__device__ __forceinline__ HDual cop_powf(const HDual& x, const HDual& y)
{
    // x^y = exp(y ln x)
    const HDual t = y * log(x);
    const double e = ::exp(t.v);
    return chain(t, e, e, e);
}
__device__ __forceinline__ HDual cop_ln(const HDual& x) { return x.v <= 0.0 ? HDual(-__builtin_huge_val()) : log(x); }
__device__ __forceinline__ HDual cop_log10(const HDual& x) { return x.v <= 0.0 ? HDual(-__builtin_huge_val()) : (1.0 / ::log(10.0)) * log(x); }
__device__ __forceinline__ HDual cop_exp(const HDual& x)
{
    const double e = ::exp(x.v);
    return chain(x, e, e, e);
}
__device__ __forceinline__ HDual cop_tan(const HDual& x)
{
    const double t = ::tan(x.v), s = 1.0 + t * t;
    return chain(x, t, s, 2.0 * t * s);
}
__device__ __forceinline__ HDual cop_asin(const HDual& x)
{
    const double s = 1.0 / ::sqrt(1.0 - x.v * x.v);
    return chain(x, ::asin(x.v), s, x.v * s * s * s);
}

}  // namespace mistark
