// smacos.hpp -- acos(x) as a fixed sequence of IEEE-754 operations (+, *, fma, correctly rounded sqrt and division), the same on
// the device and on a CPU.
//
// Why: the reference takes std::acos of clamped cosines (SM.C:782-783, 992-995) and compares the angles with thresholds
// (SM.C:923, 1391-1394, 1421-1424).  glibc's acos and the ROCm device library's differ in the last bits (the latter starts from the
// hardware's reciprocal square root / reciprocal approximations, which no CPU can reproduce), so until round 4 the angle fields of
// the engine and of the checker (oracle/) could only be compared to 1e-12, and nothing told how close a threshold comparison came
// to flipping.  With this function on BOTH sides -- the kernels use it, and the oracle evaluates it when asked to
// (orc::setAcosVariant) -- the angles are bit-identical (tests/test_gpu_parity.py: ANGLE_TOL = 0); the oracle's glibc variant stays
// the reference's arithmetic, and tests/test_oracle_acos.py counts how far the two variants' angles and decisions are apart.
//
// Method (the classic one, as in fdlibm / the ROCm device library; R = their minimax polynomial for (asin(s) - s) / s^3 in t = s^2):
//   |x| <  0.5:  acos(x) = pi/2 - (x + x t R(t)),            t = x^2
//   |x| >= 0.5:  with t = (1 - |x|) / 2, s = sqrt(t) correctly rounded, c = (t - s^2) / (2 s) its rounding error (s + c = sqrt(t) to
//                ~106 bits; t - s^2 is exact in one fma):   acos(|x|) = 2 (s + (s t R(t) + c)),   acos(-|x|) = pi - 2 (s + s t R(t))
// Error: < 1.5 ulp against a 200-bit reference over [-1, 1] (tests/test_oracle_acos.py; glibc: < 1 ulp).  No FMA CONTRACTION may
// touch this file's arithmetic (-ffp-contract=off on both sides): every fma below is written out.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)
#include "fpexact.hpp"
#define SMACOS_HD __device__ __forceinline__
#else
#define SMACOS_HD inline
#endif

namespace smacos {

SMACOS_HD double fmaX(double a, double b, double c) { return __builtin_fma(a, b, c); }
// correctly rounded square root / quotient: the device's exact fast paths give the bits of the IEEE operations (fpexact.hpp)
SMACOS_HD double sqrtX(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return smgpu::sqrtExact(x);
#else
    return __builtin_sqrt(x);
#endif
}
SMACOS_HD double divX(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(smgpu::divFastOk(a) && smgpu::divFastOk(b), 1)) return smgpu::divCore(a, smgpu::recipCore(b));
#endif
    return a / b;
}

SMACOS_HD double acosX(double x) {
    const double ax = __builtin_fabs(x);
    const bool big = ax >= 0.5;
    const double t = big ? fmaX(ax, -0.5, 0.5) : x * x;
    double p = fmaX(t, 0x1.059859fea6a70p-5, -0x1.0a5a378a05eafp-6);
    p = fmaX(t, p, 0x1.4052137024d6ap-6);
    p = fmaX(t, p, 0x1.ab3a098a70509p-8);
    p = fmaX(t, p, 0x1.8ed60a300c8d2p-7);
    p = fmaX(t, p, 0x1.c6fa84b77012bp-7);
    p = fmaX(t, p, 0x1.1c6c111dccb70p-6);
    p = fmaX(t, p, 0x1.6e89f0a0adacfp-6);
    p = fmaX(t, p, 0x1.f1c72c668963fp-6);
    p = fmaX(t, p, 0x1.6db6db41ce4bdp-5);
    p = fmaX(t, p, 0x1.333333336fd5bp-4);
    p = fmaX(t, p, 0x1.5555555555380p-3);
    const double tp = t * p;                                   // t R(t)
    if (!big) {
        const double as = fmaX(x, tp, x);                      // asin(x)
        return fmaX(0x1.dd9ad336a0500p-1, 0x1.af154eeb562d6p+0, -as);      // pi/2 as an exact product of two doubles, minus asin(x)
    }
    const double s = sqrtX(t);
    const double c = (t == 0.0) ? 0.0 : divX(fmaX(-s, s, t), 2.0 * s);
    if (x < 0.0) {
        const double w = fmaX(s, tp, s);
        return fmaX(0x1.dd9ad336a0500p+0, 0x1.af154eeb562d6p+0, -2.0 * w);  // pi (the same product, doubled) - 2 asin-part
    }
    return 2.0 * (s + fmaX(s, tp, c));
}

}  // namespace smacos
