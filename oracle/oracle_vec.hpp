// oracle_vec.hpp -- OpenFOAM Vector<double> algebra for the oracle's translation units.
// TEST INFRASTRUCTURE ONLY (see smooth_oracle.hpp).
#pragma once
#include <cmath>

#include "smooth_oracle.hpp"

namespace orc {

// VectorI.H / VectorSpaceI.H semantics: plain IEEE f64, left to right
static inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator*(double s, const Vec3& a) { return {s * a.x, s * a.y, s * a.z}; }
static inline Vec3 operator/(const Vec3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
static inline Vec3& operator+=(Vec3& a, const Vec3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline Vec3& operator/=(Vec3& a, double s) { a.x /= s; a.y /= s; a.z /= s; return a; }
static inline bool operator==(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline bool operator!=(const Vec3& a, const Vec3& b) { return !(a == b); }
static inline double dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // operator&
static inline Vec3 cross(const Vec3& a, const Vec3& b) {                                              // operator^
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double magSqr(const Vec3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline double mag(const Vec3& a) { return std::sqrt(magSqr(a)); }
static const Vec3 ZERO_VECTOR{0.0, 0.0, 0.0};          // COM.H:16
static const Vec3 UNDEF_VECTOR{GREAT, GREAT, GREAT};   // COM.H:15

static inline Vec3& operator-=(Vec3& a, const Vec3& b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
static inline Vec3 operator*(const Vec3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }

}  // namespace orc
