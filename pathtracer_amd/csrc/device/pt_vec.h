// Device-side vector helpers with GLSL evaluation order made explicit.
//
// Arithmetic contract: every kernel is compiled with -ffp-contract=off, IEEE
// division and square root (hipcc's default for HIP), and uses the built-in definitions
// of include/pt_glsl_math.h, so a value computed here is bit-identical to the same
// expression evaluated by a scalar CPU.  Operators below are component-wise and never
// reassociate; dot() sums x, y, z left to right.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_glsl_math.h"

namespace ptd {

struct vec3 {
  float x, y, z;
};

__device__ __forceinline__ vec3 mk(float x, float y, float z) { return vec3{x, y, z}; }
__device__ __forceinline__ vec3 splat(float s) { return vec3{s, s, s}; }
__device__ __forceinline__ vec3 xyz(const float4& v) { return vec3{v.x, v.y, v.z}; }

__device__ __forceinline__ vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }

__device__ __forceinline__ float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ vec3 cross(vec3 a, vec3 b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// GLSL normalize(): v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x) (both IEEE)
__device__ __forceinline__ vec3 normalize(vec3 a) {
  float inv = 1.0f / __builtin_sqrtf(dot(a, a));
  return a * inv;
}
__device__ __forceinline__ float length(vec3 a) { return __builtin_sqrtf(dot(a, a)); }

// GLSL min/max with the NaN rule this build fixes: a NaN operand is dropped
// (v_min_f32 / v_max_f32, C fminf / fmaxf).
__device__ __forceinline__ float gmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float gmax(float a, float b) { return __builtin_fmaxf(a, b); }

// column-major mat3 * vec3: c0*v.x + c1*v.y + c2*v.z per component
__device__ __forceinline__ vec3 mat3_mul(vec3 c0, vec3 c1, vec3 c2, vec3 v) {
  return {c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z,
          c0.z * v.x + c1.z * v.y + c2.z * v.z};
}

__device__ __forceinline__ vec3 pow3(vec3 a, float e) { return {pt_pow(a.x, e), pt_pow(a.y, e), pt_pow(a.z, e)}; }

__device__ __forceinline__ bool any_nan(vec3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float ubits(uint32_t u) { return __uint_as_float(u); }

}  // namespace ptd
