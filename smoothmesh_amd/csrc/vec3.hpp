// vec3.hpp -- f64 3-vector algebra with OpenFOAM Vector<double> evaluation order
// (VectorI.H: dot = x*x' + y*y' + z*z' left to right; v/s divides component-wise; no FMA:
// the translation units that include this are compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

#include "fpexact.hpp"

namespace smgpu {

#define SMGPU_HD __host__ __device__ __forceinline__

struct V3 {
    double x, y, z;
};

SMGPU_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
SMGPU_HD V3 operator+(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SMGPU_HD V3 operator-(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SMGPU_HD V3 operator*(double s, const V3& a) { return v3(s * a.x, s * a.y, s * a.z); }
SMGPU_HD V3 operator*(const V3& a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
#if defined(__HIP__)
// v / s on the device: three IEEE divisions by one denominator with the reciprocal's Newton iteration shared (fpexact.hpp),
// the plain quotients when an exponent is near the ends of the range or a numerator is zero / denormal -- the same bits
__device__ __forceinline__ V3 divExact(const V3& v, double s) {
    const unsigned hx = (unsigned)__double2hiint(v.x) & 0x7fffffffu, hy = (unsigned)__double2hiint(v.y) & 0x7fffffffu,
                   hz = (unsigned)__double2hiint(v.z) & 0x7fffffffu, hs = (unsigned)__double2hiint(s) & 0x7fffffffu;
    const unsigned lo = min(min(hx, hy), min(hz, hs)), hi = max(max(hx, hy), max(hz, hs));
    if (SMGPU_FPEXACT_FAST && __builtin_expect(lo >= ((1023u - 250u) << 20) && hi < ((1023u + 250u) << 20), 1)) {
        const Recip d = recipCore(s);
        return v3(divCore(v.x, d), divCore(v.y, d), divCore(v.z, d));
    }
    return v3(v.x / s, v.y / s, v.z / s);
}
#endif
SMGPU_HD V3 operator/(const V3& a, double s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return divExact(a, s);
#else
    return v3(a.x / s, a.y / s, a.z / s);
#endif
}
SMGPU_HD bool operator==(const V3& a, const V3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
SMGPU_HD bool operator!=(const V3& a, const V3& b) { return !(a == b); }
SMGPU_HD double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SMGPU_HD V3 cross(const V3& a, const V3& b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
SMGPU_HD double magSqr(const V3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// c ? a : b by component.  (`c ? a : b` on two V3 OBJECTS selects between their addresses: the compiler then keeps both in scratch
// memory -- 56 / 104 bytes per lane in the shared-point combine kernels until round 4.)
SMGPU_HD V3 sel3(bool c, const V3& a, const V3& b) { V3 r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z; return r; }
SMGPU_HD double mag(const V3& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return sqrtExact(magSqr(a));   // sqrt's bits, fpexact.hpp
#else
    return sqrt(magSqr(a));
#endif
}

__device__ __forceinline__ V3 ldv(const double* __restrict__ base, int i) {
    const double* p = base + 3 * (size_t)i;
    return v3(p[0], p[1], p[2]);
}
__device__ __forceinline__ void stv(double* __restrict__ base, int i, const V3& v) {
    double* p = base + 3 * (size_t)i;
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
}

// OpenFOAM doubleScalar.H constants used on the path (SM.C:259,621,1333; COM.H:15)
#define SMGPU_GREAT 1.0e+15
#define SMGPU_VSMALL 1.0e-300
#define SMGPU_ROOTVSMALL 1.0e-150
#define SMGPU_PI 3.14159265358979323846  // M_PI

}  // namespace smgpu
