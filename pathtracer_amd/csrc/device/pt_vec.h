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
// ---- the arithmetic of the SHADING half, in two builds:
//   exact (default): IEEE divide and square root, this build's own pow / sincos (pt_glsl_math.h), no FMA ever formed —
//                    bit-identical to the CPU oracle;
//   PT_FAST_ARITH  : the hardware's approximations (v_rcp_f32, v_sqrt_f32, v_rsq_f32, v_exp_f32 / v_log_f32, v_sin_f32 /
//                    v_cos_f32: ~1 ulp) and FMA contraction, the arithmetic a Vulkan driver would give main.comp. Opt-in
//                    (PT_ARITH=fast), validated by relMSE and primary-hit agreement instead of equality.
// Ray generation, the slab test and the triangle test never use these: they are written with the *_x helpers below,
// which keep contraction off in either build, so the primary-hit ids stay those of the exact build.
#ifdef PT_FAST_ARITH
__device__ __forceinline__ float sdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float ssqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float srsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float spow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }  // log2(0) = -inf -> 0; x < 0 -> NaN
__device__ __forceinline__ void ssincos(float x, float* s, float* c) {
  const float rev = x * 0.15915494309189535f;  // v_sin_f32 / v_cos_f32 take revolutions
  *s = __builtin_amdgcn_sinf(rev);
  *c = __builtin_amdgcn_cosf(rev);
}
#else
__device__ __forceinline__ float sdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float ssqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float srsq(float x) { return 1.0f / __builtin_sqrtf(x); }
__device__ __forceinline__ float spow(float x, float y) { return pt_pow(x, y); }
__device__ __forceinline__ void ssincos(float x, float* s, float* c) { pt_sincos(x, s, c); }
#endif

__device__ __forceinline__ vec3 operator/(vec3 a, float s) { return {sdiv(a.x, s), sdiv(a.y, s), sdiv(a.z, s)}; }
__device__ __forceinline__ vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }

__device__ __forceinline__ float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ vec3 cross(vec3 a, vec3 b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// GLSL normalize(): v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x) (both IEEE in the exact build)
__device__ __forceinline__ vec3 normalize(vec3 a) {
  float inv = srsq(dot(a, a));
  return a * inv;
}
__device__ __forceinline__ float length(vec3 a) { return ssqrt(dot(a, a)); }

// Exact forms for ray generation and intersection: contraction off whatever the translation unit's setting.
__device__ __forceinline__ vec3 sub_x(vec3 a, vec3 b) {
#pragma clang fp contract(off)
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ float dot_x(vec3 a, vec3 b) {
#pragma clang fp contract(off)
  const float xx = a.x * b.x, yy = a.y * b.y, zz = a.z * b.z;
  return xx + yy + zz;
}
__device__ __forceinline__ vec3 cross_x(vec3 a, vec3 b) {
#pragma clang fp contract(off)
  const float yz = a.y * b.z, zy = b.y * a.z, zx = a.z * b.x, xz = b.z * a.x, xy = a.x * b.y, yx = b.x * a.y;
  return {yz - zy, zx - xz, xy - yx};
}
__device__ __forceinline__ vec3 normalize_x(vec3 a) {
#pragma clang fp contract(off)
  const float inv = 1.0f / __builtin_sqrtf(dot_x(a, a));
  return {a.x * inv, a.y * inv, a.z * inv};
}
// a + s * b per component (the mul rounded, then the add)
__device__ __forceinline__ vec3 madd_x(vec3 a, float s, vec3 b) {
#pragma clang fp contract(off)
  const float px = s * b.x, py = s * b.y, pz = s * b.z;
  return {a.x + px, a.y + py, a.z + pz};
}

// GLSL min/max with the NaN rule this build fixes: a NaN operand is dropped
// (v_min_f32 / v_max_f32, C fminf / fmaxf).
__device__ __forceinline__ float gmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float gmax(float a, float b) { return __builtin_fmaxf(a, b); }

// column-major mat3 * vec3: c0*v.x + c1*v.y + c2*v.z per component
__device__ __forceinline__ vec3 mat3_mul(vec3 c0, vec3 c1, vec3 c2, vec3 v) {
  return {c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z,
          c0.z * v.x + c1.z * v.y + c2.z * v.z};
}

__device__ __forceinline__ vec3 pow3(vec3 a, float e) { return {spow(a.x, e), spow(a.y, e), spow(a.z, e)}; }

__device__ __forceinline__ bool any_nan(vec3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float ubits(uint32_t u) { return __uint_as_float(u); }

}  // namespace ptd
