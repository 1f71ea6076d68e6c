// fpexact.hpp -- IEEE-754 correctly rounded f64 square root and division for the geometry kernel, with the range handling the
// compiler's expansions carry on every call moved behind one test.
//
// The reference's arithmetic (OpenFOAM face / cell centres: 4 sqrt + 3 divisions per quadrilateral, 6 divisions per cell) has
// to come out bit for bit, so sqrt() and '/' stay IEEE operations.  hipcc expands them to: a scaling prologue / epilogue for
// arguments near the ends of the exponent range (v_cmp + v_cndmask + 2 v_ldexp for sqrt, 2 v_div_scale + v_div_fmas +
// v_div_fixup for a division), a class test for 0 / inf / nan, and between them a Newton iteration on v_rsq_f64 / v_rcp_f64
// (18 resp. 11 instructions).  For arguments away from the ends of the range the prologue and epilogue are the identity:
// the functions below run the SAME iteration (same instructions, same order, hence the same bits) without them when a
// two-instruction exponent test says so, and the plain operator otherwise.  Three numerators over one denominator share the
// reciprocal's iteration (5 of the 8 instructions of each division).  Checked against sqrt() and '/' on random and edge
// arguments in tests/test_gpu_fpexact.py through smgpu_selftest_fpexact.
#pragma once
#include <hip/hip_runtime.h>
#if defined(__HIP__)   // (the host-only translation units include vec3.hpp as plain C++)

// 0: every call takes the plain operator (A/B builds)
#ifndef SMGPU_FPEXACT_FAST
#define SMGPU_FPEXACT_FAST 1
#endif

namespace smgpu {

// biased exponent field of |x| within [lo, hi): one v_sub + one v_cmp on the high dword (the sign bit puts negative numbers out)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ bool expWithin(double x) {
    return SMGPU_FPEXACT_FAST && ((unsigned)__double2hiint(x) - (LO << 20)) < ((HI - LO) << 20);
}
template <unsigned LO, unsigned HI>
__device__ __forceinline__ bool absExpWithin(double x) {
    return SMGPU_FPEXACT_FAST && (((unsigned)__double2hiint(x) & 0x7fffffffu) - (LO << 20)) < ((HI - LO) << 20);
}

// the Newton iteration of the compiler's f64 sqrt for x in [2^-767, inf): no scaling, x is not 0 / inf / nan
__device__ __forceinline__ double sqrtCore(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ bool sqrtFastOk(double x) { return expWithin<1023 - 767, 2047>(x); }
__device__ __forceinline__ double sqrtExact(double x) { return __builtin_expect(sqrtFastOk(x), 1) ? sqrtCore(x) : sqrt(x); }

// Division.  Away from the range ends (here: both exponents within 2^-250 .. 2^250, so that neither the quotient nor the
// residual a - b q leaves the normal range) v_div_scale returns its argument and clears VCC, v_div_fmas is v_fma and
// v_div_fixup passes its first operand through.
struct Recip { double b, r; };
__device__ __forceinline__ bool divFastOk(double x) { return absExpWithin<1023 - 250, 1023 + 250>(x); }
__device__ __forceinline__ Recip recipCore(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    return Recip{b, r};
}
__device__ __forceinline__ double divCore(double a, const Recip& d) {
    const double q = a * d.r;
    const double e = __builtin_fma(-d.b, q, a);
    return __builtin_fma(e, d.r, q);
}

}  // namespace smgpu
#endif  // __HIP__
