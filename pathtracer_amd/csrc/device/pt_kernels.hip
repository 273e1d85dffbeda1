// Path-tracing kernels for gfx950 (wave64): the device half of the hot path, i.e. the work of
// src/renderer/shaders/main.comp re-organised from "one invocation owns a pixel for all its
// samples, one dispatch" into queues of pixel slots that are re-compacted as pixels finish:
//
//   k_init     main():588-605 for the first sample of every pixel slot
//   k_paths    (default pipeline) up to 32 path vertices of every queued slot per launch:
//              ray_scene_intersect (main.comp:263-304), then the trace() body (:517-577) with its
//              light sampling and shadow walks (:306-332, 409-501), BSDF sampling, NaN filter and
//              regeneration (main():597-612); survivors compacted into the next queue
//   k_extend / k_shade / k_shadow   (wavefront pipeline, PT_PIPELINE=wavefront) the same work as one
//              traversal launch + one shading launch per vertex, shadow walks fused into k_shade
//              (LDS-resident scenes) or deferred to k_shadow (scenes in global memory)
//   k_finalize slot order -> pixel order, float4 accumulator + primary-hit ids
//   k_resolve  main():616-619
//
// A pixel owns ONE RNG stream consumed sequentially across all its samples,
// so a pixel slot carries one path at a time and regenerates the next sample as soon as
// the current one ends; the queues hold slot indices and are compacted with ballot + mbcnt
// and one atomic per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_device.h"
#include "pt_vec.h"

using namespace ptd;

// This file is compiled three times: as it stands (the exact build: namespace ptd, every kernel); through
// pt_kernels_fast.hip, with PT_FAST_ARITH defined and -ffp-contract=fast (the fast-arithmetic build of the path
// kernel only: namespace ptd_fast, k_paths and launch_paths; see pt_vec.h for what "fast" changes); and through
// pt_kernels_rs.hip with PT_STACK_CHILD 0 (namespace ptd_rs: the traversal kernels with (record, side) stack entries, for
// scenes whose children do not fit a packed entry).
//
// PT_STACK_CHILD (scenes in global memory): 1 = a traversal-stack entry is the deferred child itself, (ref << 7 | cnt), so a
// pop is no fetch — on these scenes the rate of divergent lane loads is the bound (rocks -7 %, atrium -2.4 % frame time);
// needs every ref < 2^25 and every leaf < 128 triangles (pt_dev_scene::packed_stack, checked at upload). 0 = the entry is
// (record << 1 | side) and a pop fetches the child's 8 bytes. The choice is made at COMPILE time on purpose: as a wave-uniform
// bool tested inside the traversal loops, hipcc (ROCm 7.2) materialised the condition with v_cmp under the exec mask of one
// loop and re-used that lane mask in another loop under a different exec mask, so a push and its pop could disagree on the
// entry format (GPU memory fault, found the hard way).
#ifndef PT_STACK_CHILD
#define PT_STACK_CHILD 1
#endif
#ifdef PT_FAST_ARITH
#define PT_KNS ptk_fast
#define PT_LNS ptd_fast
#elif !PT_STACK_CHILD
#define PT_KNS ptk_exact_rs
#define PT_LNS ptd_rs
#else
#define PT_KNS ptk_exact
#define PT_LNS ptd
#endif

namespace {

constexpr float kPi = 3.1415926535f;         // main.comp:13
constexpr float kGamma = 2.2f;               // :14
constexpr float kInfinity = 3.4e+38f;        // :15
constexpr float kNormalDelta = 1e-4f;        // :16
constexpr float kPointLightRadius = 0.2f;    // :17
constexpr uint32_t kMaxTransparencySteps = 4;  // :18
constexpr uint32_t kMaxLightSteps = 8;       // :19
constexpr float kClamp = 10.0f;              // :20
constexpr float kEpsilon = 1e-6f;            // :21
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
// glslang folds these constant expressions in double precision before rounding to float
constexpr float kLightArea = static_cast<float>(4.0 * 3.1415926535 * 0.2 * 0.2);  // :450
constexpr float kFourPi = static_cast<float>(4.0 * 3.1415926535);                 // :451
constexpr float kInvGamma = static_cast<float>(1.0 / 2.2);                        // :617

// ---- per-slot counters word: light steps [0,4) | transparency steps [4,7) | primary flag 7 | sample [8,32)
__device__ __forceinline__ uint32_t ctr_light(uint32_t c) { return c & 15u; }
__device__ __forceinline__ uint32_t ctr_transp(uint32_t c) { return (c >> 4) & 7u; }
__device__ __forceinline__ bool ctr_primary(uint32_t c) { return (c >> 7) & 1u; }
__device__ __forceinline__ uint32_t ctr_sample(uint32_t c) { return c >> 8; }
__device__ __forceinline__ uint32_t ctr_pack(uint32_t light, uint32_t transp, bool primary, uint32_t sample) {
  return light | (transp << 4) | (primary ? 128u : 0u) | (sample << 8);
}

// ---------------------------------------------------------------- RNG (main.comp:135-149)
__device__ __forceinline__ uint32_t triple32(uint32_t x) {
  x ^= x >> 17; x *= 0xed5ad4bbu;
  x ^= x >> 11; x *= 0xac4c1b51u;
  x ^= x >> 15; x *= 0x31848babu;
  x ^= x >> 14;
  return x;
}
__device__ __forceinline__ float rand_float(uint32_t& state) {
  state = triple32(state);
  return static_cast<float>(state) / 4294967296.0f;  // float(~0u) == 2^32 (Q3)
}

// ---------------------------------------------------------------- wave primitives
__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {  // set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Queue compaction, two queues at once: ballot + mbcnt inside a wave, a 4-entry LDS scan
// across the block's waves, ONE returning atomic per block and queue (a single hot word
// saturates near 90 atomics/us on MI355X, so per-wave atomics would bound the kernel).
// Every thread of the block must call it (converged). Blocks have 1 to 4 waves (k_paths runs 64-thread blocks).
struct AppendSlots {
  uint32_t a, b;
};
__device__ __forceinline__ AppendSlots block_append2(bool pred_a, uint32_t* counter_a, bool pred_b, uint32_t* counter_b) {
  __shared__ uint32_t wave_count[2][4];
  __shared__ uint32_t block_base[2];
  const uint64_t mask_a = __ballot(pred_a), mask_b = __ballot(pred_b);
  const uint32_t w = wave_id(), lane = lane_id();
  if (lane == 0) {
    wave_count[0][w] = static_cast<uint32_t>(__popcll(mask_a));
    wave_count[1][w] = static_cast<uint32_t>(__popcll(mask_b));
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const uint32_t q = threadIdx.x;
    uint32_t total = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) total += wave_count[q][i];
    uint32_t* counter = q ? counter_b : counter_a;
    block_base[q] = (total && counter) ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t off_a = block_base[0], off_b = block_base[1];
  for (uint32_t i = 0; i < w; ++i) {
    off_a += wave_count[0][i];
    off_b += wave_count[1][i];
  }
  // A workgroup that strides the queue calls this once per trip: no wave may write its next
  // trip's count while a slower wave still sums this trip's (a wave with nothing to do on the
  // last, partial trip gets there at once; without this barrier 1080p frames lost ~1e-4 of their slots).
  __syncthreads();
  return AppendSlots{off_a + mbcnt(mask_a), off_b + mbcnt(mask_b)};
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

struct Tally {  // per-thread work counters, flushed once per kernel
  uint32_t rays = 0, visits = 0, tests = 0, hits = 0, taps = 0, verts = 0, paths = 0, nans = 0;
};

// Work counters without atomics: wave reduction, LDS across the block's waves, then lanes
// 0..7 of the block add the eight sums into the block's OWN row of the partial table
// (stats[kernel][blockIdx.x][word]); launches are stream-ordered, so the read-modify-write
// is race-free. The host sums the rows after the render.
__device__ __forceinline__ void flush_tally(const Tally& t, const pt_dev_state& st, int kernel) {
  __shared__ uint32_t part[4][PT_ST_WORDS];
  uint32_t v[PT_ST_WORDS] = {t.paths, t.rays, t.visits, t.tests, t.hits, t.taps, t.verts, t.nans};
  const uint32_t w = wave_id(), lane = lane_id();
#pragma unroll
  for (int i = 0; i < PT_ST_WORDS; ++i) {
    const uint32_t s = wave_sum(v[i]);
    if (lane == 0) part[w][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < PT_ST_WORDS) {
    const uint32_t waves = blockDim.x >> 6;
    uint32_t s = 0;
    for (uint32_t i = 0; i < waves; ++i) s += part[i][threadIdx.x];
    if (s) {
      unsigned long long* row = st.stats + (static_cast<size_t>(kernel) * st.stat_rows + blockIdx.x) * PT_ST_WORDS;
      row[threadIdx.x] += s;
    }
  }
  __syncthreads();  // `part` may be reused by a second flush
}

// ---------------------------------------------------------------- traversal
struct Hit {
  float t, u, v;
  uint32_t tri;
  uint32_t opq = 0;  // the accepted triangle's tag word, tri_pos[3 tri].w: non-zero = its material is provably opaque (read by shadow walks only)
};

// ray_aabb_intersect, main.comp:169-185, for BOTH children of a pair at once. inv_dir is
// hoisted out of the loop (bit-identical), and the x/y/z slab arithmetic of the left and
// right box runs in the two halves of packed fp32 instructions (v_pk_add_f32 /
// v_pk_mul_f32: a wave64 VALU op costs 4 cycles on a SIMD, packed or not, and this kernel
// is VALU-issue bound). Component-wise IEEE sub/mul, so the values equal the scalar form.
// Also applies the one extra cull the build allows itself: a box whose
// t_far is negative lies behind the ray origin; no triangle in it can pass `t > epsilon`,
// so skipping it never changes a result — only the visit/test counters, which therefore
// count after this cull (the oracle's ORC_FLAG_CULL_BEHIND mirrors it).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void ray_box_pair(vec3 o, vec3 inv, const float4& q0, const float4& q1, const float4& q2, float t_best,
                                             bool& hit_l, bool& hit_r, float& dist_l, float& dist_r, float& exit_l, float& exit_r) {
#pragma clang fp contract(off)
  // q0 = {lmin.x, rmin.x, lmin.y, rmin.y}, q1 = {lmin.z, rmin.z, lmax.x, rmax.x}, q2 = {lmax.y, rmax.y, lmax.z, rmax.z}:
  // every {left, right} operand pair of the packed instructions is an aligned register pair as loaded
  const f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f2 ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
  const f2 ax = ix * (f2{q0.x, q0.y} - ox), bx = ix * (f2{q1.z, q1.w} - ox);  // t_min.x, t_max.x of {L, R}
  const f2 ay = iy * (f2{q0.z, q0.w} - oy), by = iy * (f2{q2.x, q2.y} - oy);
  const f2 az = iz * (f2{q1.x, q1.y} - oz), bz = iz * (f2{q2.z, q2.w} - oz);
  const float near_l = gmax(gmax(gmin(ax.x, bx.x), gmin(ay.x, by.x)), gmin(az.x, bz.x));
  const float far_l = gmin(gmin(gmax(ax.x, bx.x), gmax(ay.x, by.x)), gmax(az.x, bz.x));
  const float near_r = gmax(gmax(gmin(ax.y, bx.y), gmin(ay.y, by.y)), gmin(az.y, bz.y));
  const float far_r = gmin(gmin(gmax(ax.y, bx.y), gmax(ay.y, by.y)), gmax(az.y, bz.y));
  dist_l = near_l;
  dist_r = near_r;
  exit_l = far_l;
  exit_r = far_r;
  hit_l = near_l <= far_l && !(far_l < 0.0f) && near_l < t_best;  // :179, behind-origin cull, :284
  hit_r = near_r <= far_r && !(far_r < 0.0f) && near_r < t_best;
}

// Where a traversal reads the scene from. The BVH is held as CHILD-PAIR records in
// breadth-first order (built at upload from the reference's node array, pt_context.hip):
// record p = four quads, 64 bytes:
//   {lmin.x, rmin.x, lmin.y, rmin.y | lmin.z, rmin.z, lmax.x, rmax.x | lmax.y, rmax.y, lmax.z, rmax.z |
//    left.ref, left.cnt, right.ref, right.cnt}
// (left/right interleaved: the slab test runs left and right in the two halves of packed
// fp32 instructions), where ref is the child's own pair record (inner) or its first
// triangle (leaf). IN_LDS: the scene (records + triangle positions) is small enough to be
// staged whole in the workgroup's LDS and every fetch is a ds_read_b128; otherwise the
// fetches are global loads (L1/L2/Infinity Cache/HBM).
struct Tracer {
  const float4* pairs;    // LDS or global, breadth-first
  const float4* wide;     // PT_BVH=wide4 (opt-in, global memory): 64-byte records of up to four children with 8-bit boxes; else NULL
  const float4* tri_pos;  // LDS or global, 3 x float4 per triangle
  uint32_t root_ref, root_cnt;
  uint32_t stack;         // LDS byte address of this lane's column of the traversal stack
  uint32_t stride;        // bytes from one level to the next
  uint32_t stack_end;     // scenes in global memory: LDS byte address one past this lane's last LDS row; pushes beyond it go to private memory
#ifdef PT_GUARD
  uint32_t g_pairs, g_tris, g_wide;
#endif
};

// Rows of the traversal stack beyond a lane's LDS rows (scenes in global memory may run with a short LDS stack so that
// more waves fit a CU): a private array, i.e. scratch memory, touched only by the rare ray whose stack outgrows the LDS rows.
constexpr uint32_t kDeepLevels = PT_DEEP_LEVELS;  // pt_device.h: the host sizes the short LDS stack against it

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void stack_store(uint32_t addr, uint32_t v) { *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)) = v; }
__device__ __forceinline__ uint32_t stack_load(uint32_t addr) { return *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)); }

template <bool IN_LDS>
__device__ __forceinline__ float4 fetch16(const float4* base, uint32_t index) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (IN_LDS) {  // flat LDS address -> LDS offset (its low 32 bits): a plain ds_read_b128
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f4 lds_f4;
    lds_f4* p = (lds_f4*)(uintptr_t)static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base));
    const f4 v = p[index];
    return make_float4(v.x, v.y, v.z, v.w);
  }
#endif
  return base[index];
}

// Keep 16-byte fetches from global memory WHOLE. Left to itself the compiler narrows a float4 load to the words that are used
// and re-splits it into overlapping 8-byte loads that land in aligned register pairs for the packed arithmetic: a triangle's
// three vertices became five loads (4 x dwordx2 + dwordx3). On the scenes read from global memory the kernel is bound by the
// rate of divergent lane loads, not by VALU, so one dwordx4 per vertex and a few v_mov are the better trade. The empty asm
// makes every quad a 128-bit register operand; it emits nothing and changes no value.
#ifndef PT_PIN_TRI
#define PT_PIN_TRI 1
#endif
#ifndef PT_PIN_INTERP
#define PT_PIN_INTERP 0
#endif
#ifndef PT_PIN_MAT
#define PT_PIN_MAT 0
#endif
#ifndef PT_PIN_EMIT
#define PT_PIN_EMIT 0
#endif
#ifndef PT_PIN_ATTR
#define PT_PIN_ATTR 1
#endif
#ifdef PT_GUARD
#define PT_CHECK(cond, what, a, b)                                                                       \
  if (!(cond)) {                                                                                         \
    printf("[guard] %s: %u %u (block %u thread %u)\n", what, (unsigned)(a), (unsigned)(b), blockIdx.x, threadIdx.x); \
    return false;                                                                                        \
  }
#else
#define PT_CHECK(cond, what, a, b)
#endif
typedef float pt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 as_float4(pt_f4 v) { return make_float4(v.x, v.y, v.z, v.w); }
template <bool IN_LDS, bool PIN = true>
__device__ __forceinline__ void fetch16x3(const float4* base, uint32_t index, float4& a, float4& b, float4& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (!IN_LDS && PIN) {
    const pt_f4* p = reinterpret_cast<const pt_f4*>(base) + index;
    pt_f4 x = p[0], y = p[1], z = p[2];
    asm("" : "+v"(x), "+v"(y), "+v"(z));
    a = as_float4(x);
    b = as_float4(y);
    c = as_float4(z);
    return;
  }
#endif
  a = fetch16<IN_LDS>(base, index);
  b = fetch16<IN_LDS>(base, index + 1);
  c = fetch16<IN_LDS>(base, index + 2);
}
template <bool IN_LDS, bool PIN = true>
__device__ __forceinline__ void fetch16x4(const float4* base, uint32_t index, float4& a, float4& b, float4& c, float4& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (!IN_LDS && PIN) {
    const pt_f4* p = reinterpret_cast<const pt_f4*>(base) + index;
    pt_f4 x = p[0], y = p[1], z = p[2], w = p[3];
    asm("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    a = as_float4(x);
    b = as_float4(y);
    c = as_float4(z);
    d = as_float4(w);
    return;
  }
#endif
  a = fetch16<IN_LDS>(base, index);
  b = fetch16<IN_LDS>(base, index + 1);
  c = fetch16<IN_LDS>(base, index + 2);
  d = fetch16<IN_LDS>(base, index + 3);
}

// (ref, cnt) of side 0/1 of pair record `rec`: one 8-byte fetch from the record's last quad
template <bool IN_LDS>
__device__ __forceinline__ uint2 fetch_child(const float4* pairs, uint32_t rec, uint32_t side) {
  const uint32_t index = 8u * rec + 6u + side;  // in 8-byte units
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (IN_LDS) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) u2 lds_u2;
    lds_u2* p = (lds_u2*)(uintptr_t)static_cast<uint32_t>(reinterpret_cast<uintptr_t>(pairs));
    const u2 v = p[index];
    return make_uint2(v.x, v.y);
  }
#endif
  return reinterpret_cast<const uint2*>(pairs)[index];
}

// ray_triangle_intersect, main.comp:189-217
// TAG: also note the accepted triangle's tag word (shadow walks of scenes that are not all-opaque; a template flag rather than
// dead-code elimination's business: the unconditional form moved cfg2's register allocation, -2 %)
template <bool IN_LDS, bool TAG = false>
__device__ __forceinline__ void ray_triangle(const Tracer& tr, vec3 o, vec3 d, uint32_t tri, Hit& h) {
#pragma clang fp contract(off)
  float4 v0, v1, v2;
  fetch16x3<IN_LDS, PT_PIN_TRI != 0>(tr.tri_pos, 3 * tri, v0, v1, v2);
  const vec3 p0 = xyz(v0);
  const vec3 edge1 = sub_x(xyz(v1), p0);
  const vec3 edge2 = sub_x(xyz(v2), p0);
  const vec3 ray_cross_edge2 = cross_x(d, edge2);
  const float det = dot_x(edge1, ray_cross_edge2);
  if (__builtin_fabsf(det) < kEpsilon) return;

  const float inv_det = 1.0f / det;
  const vec3 s = sub_x(o, p0);
  const float u = inv_det * dot_x(s, ray_cross_edge2);
  if (u < 0.0f || u > 1.0f) return;

  const vec3 s_cross_edge1 = cross_x(s, edge1);
  const float v = inv_det * dot_x(d, s_cross_edge1);
  if (v < 0.0f || u + v > 1.0f) return;

  const float t = inv_det * dot_x(edge2, s_cross_edge1);
  if (t > kEpsilon && t < h.t) {
    h.t = t;
    h.u = u;
    h.v = v;
    h.tri = tri;
    if (TAG) h.opq = fbits(v0.w);
  }
}

// ray_scene_intersect, main.comp:263-304. Same visiting order as the reference (near child
// first, far child deferred, leaf triangles in index order, no re-cull on pop): when both
// children are hit the near one is entered directly instead of being pushed and popped.
// The deferred far children live in an LDS stack, one column per lane
// (stack[level * stride + lane]: conflict-free ds_read/ds_write_b32); an entry names a
// child as (pair record << 1 | side), or for LDS scenes holds the child itself (ref << 16 | cnt).
// ANY_HIT (shadow rays of provably all-opaque scenes): the traversal ends at the first
// accepted triangle. The order is unchanged, so whether a hit exists is decided exactly as
// by the full closest-hit search; h then holds *a* hit, not the closest.
// Two loop shapes, same per-ray order of visits and tests:
//  - LDS-resident scenes (VALU-issue bound) run "while-while": a lane keeps descending inner
//    nodes until it stands on a leaf, and only then do the lanes of the wave that stand on
//    leaves run the triangle tests together, so the ~70-instruction triangle arm is not
//    executed on nearly every trip for the one or two lanes that need it (+3 % on cfg2).
//  - scenes read from global memory (latency bound) keep inner and leaf handling as the two
//    arms of one trip, so every lane issues its next fetch every trip (while-while was
//    13 % slower on the 1 M-triangle soup: lanes parked on a leaf have no load in flight).
//
// Which of two hit children is entered first. Closest-hit walks owe the reference its order (nearer first, the right one on a
// tie: Q14 — the tie-breaks between equally distant triangles depend on it). The answer of an ANY-HIT walk does not depend on the
// order, only its length does: it enters the child in whose box the ray stays longer, counted from the origin on
// (t_far - max(t_near, 0)), the left one on a tie — on the 5 M-triangle rocks -8.6 % of ALL node visits against nearer-first,
// -10.7 % on the 1 M-triangle soup (scripts/anyhit_order_model.py tried ten orders on the oracle, whose ORC_FLAG_ANYHIT_SPAN_ORDER
// mirrors this one so that the work counters still compare equal). Scenes read from global memory only: the few any-hit walks of an
// LDS-resident scene are short, and the extra registers cost cfg2 3 % (measured).
template <bool ANY_HIT>
__device__ __forceinline__ bool left_first(float dist_l, float dist_r, float exit_l, float exit_r) {
#pragma clang fp contract(off)
  if (ANY_HIT) return (exit_l - gmax(dist_l, 0.0f)) >= (exit_r - gmax(dist_r, 0.0f));
  return dist_l < dist_r;
}

template <bool IN_LDS, bool ANY_HIT, bool TAG = false>
__device__ __forceinline__ void traverse(const Tracer& tr, vec3 o, vec3 d, Hit& h, Tally& tally) {
  ++tally.rays;
  const vec3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  uint32_t sp = tr.stack;      // LDS address of the next free stack entry
  uint32_t ref = tr.root_ref;  // root: never box-tested (Q13)
  uint32_t cnt = tr.root_cnt;
  bool live = true;
  // pop the next deferred child into (ref, cnt), or end the traversal
  auto pop = [&]() {
    if (sp == tr.stack) {
      live = false;
      return;
    }
    sp -= tr.stride;
    const uint32_t e = stack_load(sp);
    if (IN_LDS) {  // the entry IS the child, (ref << 16 | cnt): both are below 65536 in a scene that fits the LDS
      ref = e >> 16;
      cnt = e & 0xFFFFu;
      return;
    }
    const uint2 c = fetch_child<IN_LDS>(tr.pairs, e >> 1, e & 1u);
    ref = c.x;
    cnt = c.y;
  };
  auto inner = [&]() {
    ++tally.visits;
    const float4 q0 = fetch16<IN_LDS>(tr.pairs, 4 * ref), q1 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 1);
    const float4 q2 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 2), q3 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 3);
    float dist_l, dist_r, exit_l, exit_r;
    bool hit_l, hit_r;
    ray_box_pair(o, inv, q0, q1, q2, h.t, hit_l, hit_r, dist_l, dist_r, exit_l, exit_r);
    if (hit_l || hit_r) {
      // (the comparison written here, not behind left_first<false>: as a call hipcc materialised the bools with v_cndmask / v_cmp, five more
      // VALU instructions per inner trip, cfg2 -1.9 %)
      const bool go_left = hit_l && (!hit_r || dist_l < dist_r);  // equal distances: right is near (Q14); LDS scenes: any-hit walks too
      if (hit_l && hit_r) {
        // the far side: the child itself for LDS scenes (a pop then needs no fetch: +0.4 % on cfg2), else (record, side)
        if (IN_LDS) stack_store(sp, go_left ? (fbits(q3.z) << 16) | fbits(q3.w) : (fbits(q3.x) << 16) | fbits(q3.y));
        else stack_store(sp, (ref << 1) | (go_left ? 1u : 0u));
        sp += tr.stride;
      }
      ref = go_left ? fbits(q3.x) : fbits(q3.z);
      cnt = go_left ? fbits(q3.y) : fbits(q3.w);
    } else {
      pop();
    }
  };
  auto leaf = [&]() -> bool {  // true: an any-hit traversal is decided
    if (ANY_HIT) {
      for (uint32_t i = 0; i < cnt; ++i) {
        ++tally.tests;
        ray_triangle<IN_LDS, TAG>(tr, o, d, ref + i, h);
        if (h.tri != kNoHit) return true;
      }
    } else {
      tally.tests += cnt;
      for (uint32_t i = 0; i < cnt; ++i) ray_triangle<IN_LDS, TAG>(tr, o, d, ref + i, h);
    }
    pop();
    return false;
  };
  if (IN_LDS) {
    while (live) {
      while (live && cnt == 0) inner();
      while (live && cnt > 0)
        if (leaf()) return;
    }
  } else {
    uint32_t deep[kDeepLevels];  // rows beyond the lane's LDS rows (private memory; nd counts them)
    uint32_t nd = 0;
    for (;;) {
#ifdef PT_GUARD
      if (cnt > 0 ? (ref + cnt > tr.g_tris) : (ref >= tr.g_pairs)) {
        printf("[guard] traverse: ref %u cnt %u nd %u (block %u thread %u)\n", ref, cnt, nd, blockIdx.x, threadIdx.x);
        return;
      }
      if (nd > kDeepLevels || sp < tr.stack || sp > tr.stack_end) {
        printf("[guard] traverse stack: sp %u nd %u (block %u thread %u)\n", sp, nd, blockIdx.x, threadIdx.x);
        return;
      }
#endif
      if (cnt > 0) {
        if (ANY_HIT) {
          for (uint32_t i = 0; i < cnt; ++i) {
            ++tally.tests;
            ray_triangle<IN_LDS, TAG>(tr, o, d, ref + i, h);
            if (h.tri != kNoHit) return;
          }
        } else {
          tally.tests += cnt;
          for (uint32_t i = 0; i < cnt; ++i) ray_triangle<IN_LDS, TAG>(tr, o, d, ref + i, h);
        }
      } else {
        ++tally.visits;
        const float4 q0 = fetch16<IN_LDS>(tr.pairs, 4 * ref), q1 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 1);
        const float4 q2 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 2), q3 = fetch16<IN_LDS>(tr.pairs, 4 * ref + 3);
        float dist_l, dist_r, exit_l, exit_r;
        bool hit_l, hit_r;
        ray_box_pair(o, inv, q0, q1, q2, h.t, hit_l, hit_r, dist_l, dist_r, exit_l, exit_r);
        if (hit_l || hit_r) {
          bool go_left;
          if constexpr (ANY_HIT) go_left = hit_l && (!hit_r || left_first<true>(dist_l, dist_r, exit_l, exit_r));
          else go_left = hit_l && (!hit_r || dist_l < dist_r);
          if (hit_l && hit_r) {
#if PT_STACK_CHILD  // the far side: the child itself, or (record, side)
            const uint32_t e = ((go_left ? fbits(q3.z) : fbits(q3.x)) << 7) | (go_left ? fbits(q3.w) : fbits(q3.y));
#else
            const uint32_t e = (ref << 1) | (go_left ? 1u : 0u);
#endif
            if (sp != tr.stack_end) {
              stack_store(sp, e);
              sp += tr.stride;
            } else {
              deep[nd++] = e;
            }
          }
          ref = go_left ? fbits(q3.x) : fbits(q3.z);
          cnt = go_left ? fbits(q3.y) : fbits(q3.w);
          continue;
        }
      }
      uint32_t e;
      if (nd != 0) {
        e = deep[--nd];
      } else {
        if (sp == tr.stack) break;
        sp -= tr.stride;
        e = stack_load(sp);
      }
#if PT_STACK_CHILD
      ref = e >> 7;
      cnt = e & 127u;
#else
      const uint2 c = fetch_child<IN_LDS>(tr.pairs, e >> 1, e & 1u);
      ref = c.x;
      cnt = c.y;
#endif
    }
  }
}

// Workgroup prologue of the traversal kernels: carve the dynamic LDS into
// [stack | pair records | triangles] and, for IN_LDS, stage the scene.
template <bool IN_LDS>
__device__ __forceinline__ Tracer make_tracer(const pt_dev_scene& sc, uint32_t* lds) {
  Tracer tr;
  tr.pairs = sc.pairs;
  tr.wide = sc.wide;
  tr.tri_pos = sc.tri_pos;
  if (IN_LDS) {
    float4* lds_pair = reinterpret_cast<float4*>(lds + static_cast<size_t>(blockDim.x) * sc.stack_entries);
    float4* lds_tri = lds_pair + 4 * static_cast<size_t>(sc.pair_count);
    for (uint32_t i = threadIdx.x; i < 4u * sc.pair_count; i += blockDim.x) lds_pair[i] = sc.pairs[i];
    for (uint32_t i = threadIdx.x; i < 3u * sc.triangle_count; i += blockDim.x) lds_tri[i] = sc.tri_pos[i];
    __syncthreads();
    tr.pairs = lds_pair;
    tr.tri_pos = lds_tri;
  }
  tr.root_ref = sc.root_ref;
  tr.root_cnt = sc.root_cnt;
  tr.stack = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds + threadIdx.x));  // flat LDS address -> LDS offset (its low 32 bits)
  tr.stride = 4u * blockDim.x;
  tr.stack_end = 0xFFFFFFFFu;
#ifdef PT_GUARD
  tr.g_pairs = sc.pair_count;
  tr.g_tris = sc.triangle_count;
  tr.g_wide = sc.wide_count;
#endif
  if (!IN_LDS) tr.stack_end = tr.stack + (sc.lds_stack_levels ? sc.lds_stack_levels : sc.stack_entries) * tr.stride;
  return tr;
}

// What shading reads besides the texel pool. For LDS-resident scenes these arrays are staged
// next to the traversal data: with them in global memory a vertex cost four more levels of
// dependent loads (attributes -> material -> emitter index -> emitter position) that the
// three waves per SIMD of the shading kernel could not hide (48 % VALU busy, profiles/r01c).
struct ShadeData {
  const float4* tri_pos;
  const float4* tri_attr;
  const float4* materials;
  const float4* dlights;
  const float4* plights;
  const uint32_t* emissive;
};

__host__ __device__ __forceinline__ uint32_t shade_lds_float4s(const pt_dev_scene& sc) {
  return 4u * sc.triangle_count + 4u * sc.material_count + 2u * (sc.directional_light_count + sc.point_light_count) +
         (sc.emissive_count + 3u) / 4u;
}

template <bool IN_LDS>
__device__ __forceinline__ ShadeData make_shade_data(const pt_dev_scene& sc, const Tracer& tr, uint32_t* lds) {
  ShadeData sd{sc.tri_pos, sc.tri_attr, sc.materials, sc.directional_lights, sc.point_lights, sc.emissive_triangles};
  if (IN_LDS) {  // after make_tracer's [stack | pairs | triangles]
    float4* attr = reinterpret_cast<float4*>(lds + static_cast<size_t>(blockDim.x) * sc.stack_entries) +
                   4 * static_cast<size_t>(sc.pair_count) + 3 * static_cast<size_t>(sc.triangle_count);
    float4* mats = attr + 4 * static_cast<size_t>(sc.triangle_count);
    float4* dl = mats + 4 * static_cast<size_t>(sc.material_count);
    float4* pl = dl + 2 * static_cast<size_t>(sc.directional_light_count);
    uint32_t* em = reinterpret_cast<uint32_t*>(pl + 2 * static_cast<size_t>(sc.point_light_count));
    for (uint32_t i = threadIdx.x; i < 4u * sc.triangle_count; i += blockDim.x) attr[i] = sc.tri_attr[i];
    for (uint32_t i = threadIdx.x; i < 4u * sc.material_count; i += blockDim.x) mats[i] = sc.materials[i];
    for (uint32_t i = threadIdx.x; i < 2u * sc.directional_light_count; i += blockDim.x) dl[i] = sc.directional_lights[i];
    for (uint32_t i = threadIdx.x; i < 2u * sc.point_light_count; i += blockDim.x) pl[i] = sc.point_lights[i];
    for (uint32_t i = threadIdx.x; i < sc.emissive_count; i += blockDim.x) em[i] = sc.emissive_triangles[i];
    __syncthreads();
    sd = ShadeData{tr.tri_pos, attr, mats, dl, pl, em};
  }
  return sd;
}

template <bool IN_LDS>
__device__ __forceinline__ uint32_t fetch32(const uint32_t* base, uint32_t index) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (IN_LDS) {
    typedef const __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* p = (lds_u32*)(uintptr_t)static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base));
    return p[index];
  }
#endif
  return base[index];
}

// ---------------------------------------------------------------- scene fetch
struct TriAttr {  // 64-byte attribute record: normals, uvs, material of one triangle
  vec3 n0, n1, n2;
  float u0, v0, u1, v1, u2, v2;
  uint32_t mat;
};

template <bool IN_LDS>
__device__ __forceinline__ TriAttr load_attr(const float4* tri_attr, uint32_t tri) {
  float4 a, b, c, e;
  fetch16x4<IN_LDS, PT_PIN_ATTR != 0>(tri_attr, 4 * tri, a, b, c, e);
  TriAttr r;
  r.n0 = mk(a.x, a.y, a.z);
  r.n1 = mk(a.w, b.x, b.y);
  r.n2 = mk(b.z, b.w, c.x);
  r.u0 = c.y; r.v0 = c.z; r.u1 = c.w;
  r.v1 = e.x; r.u2 = e.y; r.v2 = e.z;
  r.mat = fbits(e.w);
  return r;
}

struct Material {  // main.comp:42-58
  float4 base_color;
  int base_tex;
  float metallic, roughness;
  int mr_tex;
  float occlusion_strength;
  int occlusion_tex;
  vec3 emissive;
  int emissive_tex;
};

template <bool IN_LDS>
__device__ __forceinline__ Material load_material(const float4* mats, uint32_t idx) {
  float4 a, b, c, e;
  fetch16x4<IN_LDS, PT_PIN_MAT != 0>(mats, 4 * idx, a, b, c, e);
  Material m;
  m.base_color = a;
  m.base_tex = static_cast<int>(fbits(b.x));
  m.metallic = b.y;
  m.roughness = b.z;
  m.mr_tex = static_cast<int>(fbits(b.w));
  m.occlusion_strength = c.z;
  m.occlusion_tex = static_cast<int>(fbits(c.w));
  m.emissive = mk(e.x, e.y, e.z);
  m.emissive_tex = static_cast<int>(fbits(e.w));
  return m;
}

// texture(): R8G8B8A8_UNORM, single mip, LOD 0 -> magnification filter; unnormalised
// coordinate s*W - 0.5 (linear) or floor(s*W) (nearest); wrap on integer texel coordinates
// (render_job.cc:246-252, 268-292, 889-915). Same definition as the CPU oracle.
__device__ __forceinline__ int wrap_coord(int i, int n, uint32_t mode) {
  if (mode == 2u) {  // repeat
    int m = i % n;
    return m < 0 ? m + n : m;
  }
  if (mode == 1u) {  // mirrored repeat
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
  }
  return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);  // clamp to edge
}

__device__ __forceinline__ float4 texel(const uint32_t* __restrict__ texels, uint32_t base, int w, int x, int y) {
  const uint32_t p = texels[base + static_cast<uint32_t>(y) * static_cast<uint32_t>(w) + static_cast<uint32_t>(x)];
  // UNORM decode with the IEEE divide in BOTH builds: main.comp branches on alpha == 1.0 / alpha < 1.0 (:325, :542), and with the fast
  // build's x * rcp(255) a 255 byte came out as 0.99999994 — every textured wall a pass-through surface (cfg3: relMSE 0.039, found by
  // bench.py's fast_arith leg once the headline moved there)
  return make_float4(static_cast<float>(p & 255u) / 255.0f, static_cast<float>((p >> 8) & 255u) / 255.0f,
                     static_cast<float>((p >> 16) & 255u) / 255.0f, static_cast<float>(p >> 24) / 255.0f);
}

// sample_texture, main.comp:221-224
__device__ __forceinline__ float4 sample_texture(const pt_dev_scene& sc, int index, float s, float t, Tally& tally) {
  if (index == -1) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
  ++tally.taps;
  const uint4 rec = sc.textures[index];  // texel offset, width, height, mag | wrap_s << 2 | wrap_t << 4
  const int W = static_cast<int>(rec.y), H = static_cast<int>(rec.z);
  const uint32_t wrap_s = (rec.w >> 2) & 3u, wrap_t = (rec.w >> 4) & 3u;
  if ((rec.w & 3u) == 0u) {  // nearest
    const int i = static_cast<int>(__builtin_floorf(s * static_cast<float>(W)));
    const int j = static_cast<int>(__builtin_floorf(t * static_cast<float>(H)));
    return texel(sc.texels, rec.x, W, wrap_coord(i, W, wrap_s), wrap_coord(j, H, wrap_t));
  }
  const float u = s * static_cast<float>(W) - 0.5f, v = t * static_cast<float>(H) - 0.5f;
  const float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
  const float a = u - fu, b = v - fv;
  const int i0 = static_cast<int>(fu), j0 = static_cast<int>(fv);
  const int x0 = wrap_coord(i0, W, wrap_s), x1 = wrap_coord(i0 + 1, W, wrap_s);
  const int y0 = wrap_coord(j0, H, wrap_t), y1 = wrap_coord(j0 + 1, H, wrap_t);
  const float4 t00 = texel(sc.texels, rec.x, W, x0, y0), t10 = texel(sc.texels, rec.x, W, x1, y0);
  const float4 t01 = texel(sc.texels, rec.x, W, x0, y1), t11 = texel(sc.texels, rec.x, W, x1, y1);
  // lerp(b, lerp(a, t00, t10), lerp(a, t01, t11)) with lerp(w, x, y) = x + w * (y - x): constant for four equal texels, as every
  // hardware filter and llvmpipe are — an all-opaque texture's alpha is exactly 1.0 (same text as the oracle's sample_texture)
  float4 r;
  { const float lo = t00.x + a * (t10.x - t00.x), hi = t01.x + a * (t11.x - t01.x); r.x = lo + b * (hi - lo); }
  { const float lo = t00.y + a * (t10.y - t00.y), hi = t01.y + a * (t11.y - t01.y); r.y = lo + b * (hi - lo); }
  { const float lo = t00.z + a * (t10.z - t00.z), hi = t01.z + a * (t11.z - t01.z); r.z = lo + b * (hi - lo); }
  { const float lo = t00.w + a * (t10.w - t00.w), hi = t01.w + a * (t11.w - t01.w); r.w = lo + b * (hi - lo); }
  return r;
}

// sRGB decode of a sampled colour; an untextured lookup is vec4(1) and pow(1, 2.2) == 1 exactly
__device__ __forceinline__ vec3 decode_srgb(int tex_index, const float4& s) {
  if (tex_index == -1) return splat(1.0f);
  return pow3(mk(s.x, s.y, s.z), kGamma);
}

// get_interpolated_data without the normal (the shadow walk never reads it), main.comp:226-240
template <bool IN_LDS>
__device__ __forceinline__ void interpolate_pos_uv(const ShadeData& sd, uint32_t tri, vec3 wuv, vec3& pos,
                                                   float& tu, float& tv, uint32_t& mat, vec3* normal, Tally& tally) {
  ++tally.hits;
  float4 q0, q1, q2;
  fetch16x3<IN_LDS, PT_PIN_INTERP != 0>(sd.tri_pos, 3 * tri, q0, q1, q2);
  const TriAttr at = load_attr<IN_LDS>(sd.tri_attr, tri);
  pos = mat3_mul(xyz(q0), xyz(q1), xyz(q2), wuv);
  if (normal) *normal = normalize(mat3_mul(at.n0, at.n1, at.n2, wuv));
  tu = at.u0 * wuv.x + at.u1 * wuv.y + at.u2 * wuv.z;
  tv = at.v0 * wuv.x + at.v1 * wuv.y + at.v2 * wuv.z;
  mat = at.mat;
}

// ---------------------------------------------------------------- BRDF (main.comp:336-366)
__device__ __forceinline__ float distribution_ggx(float n_dot_h, float a2) {
  if (a2 < kEpsilon && n_dot_h > 1.0f - kEpsilon) return 1.0f;
  float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
  denom = kPi * denom * denom;
  return sdiv(a2, denom);
}

__device__ __forceinline__ float geometry_smith1(vec3 n, vec3 h, vec3 x, float a2) {
  if (dot(h, x) <= 0.0f) return 0.0f;
  const float n_dot_x = dot(n, x);
  const float denom = __builtin_fabsf(n_dot_x) + ssqrt(a2 + (1.0f - a2) * n_dot_x * n_dot_x);
  return sdiv(1.0f, denom);
}

__device__ __forceinline__ vec3 microfacet_brdf(vec3 n, vec3 v, vec3 l, float metal, float a2, vec3 base_color) {
  const vec3 h = normalize(l + v);
  const float n_dot_h = gmax(dot(n, h), kEpsilon);
  const vec3 f0 = splat(0.04f) * (1.0f - metal) + base_color * metal;  // mix()
  const vec3 f = f0 + (splat(1.0f) - f0) * spow(1.0f - __builtin_fabsf(dot(v, h)), 5.0f);
  const vec3 diffuse = (splat(1.0f) - f) * (1.0f - metal) * base_color / kPi;
  // geometry_smith2(n, h, l, v, a2) at the call site: G1(l) * G1(v) with the roles as written (:353, :364)
  const float g2 = geometry_smith1(n, h, v, a2) * geometry_smith1(n, h, l, a2);
  const vec3 specular = f * distribution_ggx(n_dot_h, a2) * g2;
  return diffuse + specular;
}

// ---------------------------------------------------------------- sampling (main.comp:151-165, 370-405)
struct Basis {
  vec3 u, v, w;
};
__device__ __forceinline__ Basis orthogonal_matrix(vec3 w) {
  const vec3 a = (__builtin_fabsf(w.x) > 0.9f) ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
  Basis b;
  b.v = normalize(cross(w, a));
  b.u = cross(w, b.v);
  b.w = w;
  return b;
}
__device__ __forceinline__ vec3 to_world(const Basis& b, vec3 p) { return mat3_mul(b.u, b.v, b.w, p); }

__device__ __forceinline__ vec3 uniform_hemisphere_sampling(vec3 n, uint32_t& rng) {
  const float phi = 2.0f * kPi * rand_float(rng);
  const float cos_theta = 1.0f - rand_float(rng);
  const float sin_theta = ssqrt(1.0f - cos_theta * cos_theta);
  float sp, cp;
  ssincos(phi, &sp, &cp);
  return to_world(orthogonal_matrix(n), mk(cp * sin_theta, sp * sin_theta, cos_theta));
}

__device__ __forceinline__ vec3 diffuse_sample(vec3 n, const Basis& basis, uint32_t& rng, float& pdf) {
  const float phi = 2.0f * kPi * rand_float(rng);
  const float rnd = rand_float(rng);
  float sp, cp;
  ssincos(phi, &sp, &cp);
  const float sr = ssqrt(rnd);
  const vec3 l = to_world(basis, mk(cp * sr, sp * sr, ssqrt(1.0f - rnd)));
  pdf = sdiv(gmax(dot(n, l), 0.0f), kPi);
  return l;
}

__device__ __forceinline__ vec3 microfacet_sample(const Basis& basis, vec3 v, float a2, uint32_t& rng, float& pdf) {
  const float phi = 2.0f * kPi * rand_float(rng);
  const float rnd = rand_float(rng);
  const float cos2_theta = gmin(gmax(sdiv(1.0f - rnd, rnd * (a2 - 1.0f) + 1.0f), 0.0f), 1.0f);
  const float cos_theta = ssqrt(cos2_theta);
  const float sin_theta = ssqrt(1.0f - cos2_theta);
  float sp, cp;
  ssincos(phi, &sp, &cp);
  const vec3 h = normalize(to_world(basis, mk(cp * sin_theta, sp * sin_theta, cos_theta)));
  const vec3 i = -v;
  const vec3 l = i - h * (2.0f * dot(h, i));  // reflect(-v, h)
  pdf = sdiv(distribution_ggx(cos_theta, a2) * cos_theta, 4.0f * __builtin_fabsf(dot(h, l)));
  return l;
}

// ---------------------------------------------------------------- sample bookkeeping (main():597-612)
__device__ __forceinline__ vec3 clamp_radiance(vec3 c) {
  return mk(gmin(gmax(c.x, 0.0f), kClamp), gmin(gmax(c.y, 0.0f), kClamp), gmin(gmax(c.z, 0.0f), kClamp));
}

// End of trace() (main.comp:580-582) + NaN filter (:607-612, tested before the clamp: Q12).
// accum.xyz = sum of valid samples, accum.w = number of dropped (NaN) samples.
__device__ __forceinline__ void finish_sample(vec3 color, vec3 indirect, vec3 bg_linear, float4& accum, Tally& tally) {
  const vec3 c = color + indirect * bg_linear;
  if (any_nan(c)) {
    accum.w += 1.0f;
    ++tally.nans;
  } else {
    const vec3 cc = clamp_radiance(c);
    accum.x += cc.x;
    accum.y += cc.y;
    accum.z += cc.z;
  }
}

// slot <-> pixel: slots are 8x8 pixel tiles (64 consecutive slots = one wave = one tile)
// (a slot index is local to the band being rendered; f.slot_base is the band's first slot)
__device__ __forceinline__ bool slot_to_pixel(const pt_dev_frame& f, uint32_t local_slot, uint32_t& x, uint32_t& y) {
  const uint32_t slot = local_slot + f.slot_base;
  const uint32_t tile = slot >> 6, in = slot & 63u;
  const uint32_t ty = tile / f.tiles_x, tx = tile - ty * f.tiles_x;
  x = tx * 8u + (in & 7u);
  y = ty * 8u + (in >> 3);
  return x < f.pc.resolution_x && y < f.pc.resolution_y;
}

// Camera ray of stratum `sample` (main():599-605). Draws the x jitter first, then y.
__device__ __forceinline__ void camera_ray(const pt_dev_frame& f, uint32_t x, uint32_t y, uint32_t sample, uint32_t& rng,
                                           vec3& origin, vec3& dir) {
#pragma clang fp contract(off)
  const uint32_t n = f.sqrt_samples;
  const uint32_t sub_i = sample / n, sub_j = sample - sub_i * n;
  const float fn = static_cast<float>(n);
  const float jx = (static_cast<float>(sub_i) + rand_float(rng)) / fn;
  const float jy = (static_cast<float>(sub_j) + rand_float(rng)) / fn;
  const vec3 corner = mk(f.pc.corner_pixel_pos[0], f.pc.corner_pixel_pos[1], f.pc.corner_pixel_pos[2]);
  const vec3 du = mk(f.pc.pixel_delta_u[0], f.pc.pixel_delta_u[1], f.pc.pixel_delta_u[2]);
  const vec3 dv = mk(f.pc.pixel_delta_v[0], f.pc.pixel_delta_v[1], f.pc.pixel_delta_v[2]);
  // (exact helpers: the primary rays, and with them the primary-hit ids, are the same in the exact and the fast build)
  const vec3 pixel_center = madd_x(madd_x(corner, static_cast<float>(x) + jx, du), static_cast<float>(y) + jy, dv);
  origin = mk(f.pc.camera_position[0], f.pc.camera_position[1], f.pc.camera_position[2]);
  dir = normalize_x(sub_x(pixel_center, origin));
}

}  // namespace

// =========================================================================== kernels

#ifndef PT_FAST_ARITH
namespace PT_KNS {  // kernels live in the build's own namespace: the same template compiled twice must not share a symbol
// First sample of every slot; fills queue 0 with the live slots (compacted: padding slots
// of partial tiles never enter a queue).
__global__ __launch_bounds__(256) void k_init(pt_dev_frame f, pt_dev_state st) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = 0, y = 0;
  bool live = slot < f.slot_count && slot_to_pixel(f, slot, x, y) && f.sample_begin < f.sample_end;
  Tally tally;
  if (slot < f.slot_count) {
    st.accum[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    st.primary_id[slot] = kNoHit;
  }
  if (live) {
    const uint32_t index = y * f.pc.resolution_x + x;
    uint32_t rng = index ^ f.pc.seed;  // main.comp:592
    if (f.num_ranks > 1) rng = triple32(index ^ f.pc.seed ^ (0x9E3779B9u * (f.rank + 1u)));
    vec3 o, d;
    camera_ray(f, x, y, f.sample_begin, rng, o, d);
    ++tally.paths;
    st.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.0f);
    st.ray_d[slot] = make_float4(d.x, d.y, d.z, 0.0f);
    st.path_a[slot] = make_float4(1.0f, 1.0f, 1.0f, ubits(rng));
    st.path_b[slot] = make_float4(0.0f, 0.0f, 0.0f, ubits(ctr_pack(0, 0, true, f.sample_begin)));
  }
  const AppendSlots at = block_append2(live, &st.counters[PT_CNT_QUEUE0], false, nullptr);
  if (live) st.queue[0][at.a] = slot;
  flush_tally(tally, st, PT_K_INIT);
}

// Closest hit for every queued ray. Persistent workgroups: the grid is a few blocks per CU
// and each block walks the queue in strides, so the LDS scene copy is staged once per
// block, not once per 256 rays.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void k_extend(pt_dev_scene sc, pt_dev_state st, uint32_t parity) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = st.counters[PT_CNT_QUEUE0 + parity];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // recycle the counters the previous iteration consumed
    st.counters[PT_CNT_QUEUE0 + (parity ^ 1u)] = 0;
    st.counters[PT_CNT_SHADOW] = 0;
  }
  Tally tally;
  if (blockIdx.x * blockDim.x < n) {  // block-uniform
    const Tracer tr = make_tracer<IN_LDS>(sc, lds_dyn);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const uint32_t slot = st.queue[parity][i];
      const float4 d4 = st.ray_d[slot];
      if (d4.w != 0.0f) continue;  // retired since the queue was last compacted
      const float4 o4 = st.ray_o[slot];
      Hit h{kInfinity, 0.0f, 0.0f, kNoHit};
      traverse<IN_LDS, false>(tr, xyz(o4), xyz(d4), h, tally);
      st.hit[slot] = make_uint4(h.tri, fbits(h.u), fbits(h.v), fbits(h.t));
    }
  }
  flush_tally(tally, st, PT_K_EXTEND);
}

}  // namespace PT_KNS
#endif  // !PT_FAST_ARITH

// ray_scene_intersect_with_transparency, main.comp:306-332. Returns the transparency and
// leaves the last traversal's result in h.
// RESUME (the merged walk of k_paths_merged): the first step's traversal has been done by the caller and h holds its hit.
template <bool IN_LDS, bool ALL_OPAQUE, bool TEXTURED = true, bool RESUME = false>
__device__ __forceinline__ float shadow_walk(const pt_dev_scene& sc, const ShadeData& sd, const Tracer& tr, vec3 origin, vec3 dir,
                                             Hit& h, Tally& tally) {
  float transparency = 1.0f;
  if (ALL_OPAQUE) {  // every closest hit has alpha == 1 and returns 0 (main.comp:325): any hit decides
    h.t = kInfinity;
    h.tri = kNoHit;
    traverse<IN_LDS, true>(tr, origin, dir, h, tally);
    return h.tri == kNoHit ? 1.0f : 0.0f;
  }
  for (uint32_t step = 0; step < kMaxTransparencySteps; ++step) {
    if (!(RESUME && step == 0)) {
      h.t = kInfinity;
      h.tri = kNoHit;
      traverse<IN_LDS, false, true>(tr, origin, dir, h, tally);
    }
    if (h.tri == kNoHit) break;
    // The closest hit lies on a triangle whose material is provably opaque (tagged at upload: base_color_factor.a == 1 and no
    // base-colour texture or one whose alpha bytes are all 255): alpha == 1.0 whatever the uv, so the walk returns 0 (main.comp:325)
    // without the 112-byte attribute fetch, the material and the texture tap (the oracle's ORC_FLAG_OPAQUE_TRIANGLES mirrors it).
    if (h.opq != 0u) return 0.0f;
    const vec3 wuv = mk(1.0f - h.u - h.v, h.u, h.v);
    vec3 pos;
    float tu, tv;
    uint32_t mat_idx;
    interpolate_pos_uv<IN_LDS>(sd, h.tri, wuv, pos, tu, tv, mat_idx, nullptr, tally);
    const float4 base = fetch16<IN_LDS>(sd.materials, 4 * mat_idx);
    const int base_tex = TEXTURED ? static_cast<int>(fbits(fetch16<IN_LDS>(sd.materials, 4 * mat_idx + 1).x)) : -1;
    const float alpha = base.w * sample_texture(sc, base_tex, tu, tv, tally).w;
    if (alpha == 1.0f) return 0.0f;  // any opaque closest hit, at any distance (Q5)
    transparency *= 1.0f - alpha;
    origin = pos + kNormalDelta * dir;
  }
  return transparency;
}

// One path vertex: the trace() loop body (main.comp:517-577) for an extended ray — miss
// handling, interpolation, material, light sampling WITH its shadow walks, BSDF sampling —
// then the sample bookkeeping and path regeneration of main():597-612. Returns whether the
// slot continues with the ray (next_o, next_d). Shared by k_shade (one vertex per launch) and
// k_paths (up to max_vertices vertices of a slot per launch).
// The shadow walks run here rather than in a kernel of their own: a separate shadow kernel
// needed a 112-byte request per vertex and light written and read back through HBM, and
// had to evaluate the BRDF, pdf and emitter data of every light sample up front although
// they only matter for the few samples the walk lets through (measured: DESIGN.md).
struct PathRegs {
  vec3 indirect, color;
  uint32_t rng, light_steps, transp_steps, sample;
};

// ---- k_paths_merged: a vertex's light samples wait for the NEXT traversal phase (DEFER) ----------------------------------------
// The lanes of a wave walk in lock step per phase, and a phase lasts as long as its longest lane: closest hit of vertex k, then the
// shadow walk(s) of vertex k. Merged, a lane that has finished its shadow walk goes straight on with its next ray, and the wave pays
// max over lanes of (shadow + closest) instead of max(shadow) + max(closest): scripts/merge_model.py replays k_paths' schedule on the
// oracle's trip log: -21 % wave trips on the 5 M-triangle rocks, -23 % on the atrium, -18 % on the 1 M-triangle soup.
// So shade_vertex<DEFER> walks nothing: every light sample it could not decide with a one-triangle test is left in `Pending` (its
// direction in registers, the factors of its contribution in the lane's LDS stash: the SPLIT pipeline's request, kept on chip) and the
// vertex's `color +=` (main.comp:545 / :552) is done by resolve_pending after the next merged walk. Same expressions in the same
// order as the fused form (the SPLIT form has been parity-tested since round 1), so the bits do not change.
enum : uint32_t {
  PEND_S0 = 1u, PEND_S1 = 2u,        // slot j holds a light sample that needs a walk
  PEND_AREA0 = 4u, PEND_AREA1 = 8u,  // slot j is the area sample (its contribution replaces area_light; others add to punctual)
  PEND_TAP0 = 16u, PEND_TAP1 = 32u,  // slot j's emitter has an emissive texture: the reference taps it only if the sample is visible
  PEND_SHADED = 64u,                 // a colour update is due
  PEND_ENDED = 128u                  // the path ended at that vertex: finish the sample after the update
};
struct Pending {
  vec3 origin;        // the vertex's shadow origin (pos + NORMAL_DELTA * n)
  vec3 l0, l1;        // directions of the (at most two) light samples
  vec3 emission, m;   // color += ((emission + punctual + area_light) * af) * m
  float af;
  uint32_t flags = 0;
};
struct Stash {        // LDS byte addresses of this lane's stash: per slot two quads {a_factor.xyz, correction | brdf.xyz, n_dot_l} and a pair {pdf, limit}
  uint32_t quad, quad_stride, pair, pair_stride;
};
typedef float pt_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pt_f4 lds_pt_f4;
typedef __attribute__((address_space(3))) pt_f2 lds_pt_f2;
__device__ __forceinline__ void lds_store16(uint32_t addr, float a, float b, float c, float d) {
  *reinterpret_cast<lds_pt_f4*>(static_cast<uintptr_t>(addr)) = pt_f4{a, b, c, d};
}
__device__ __forceinline__ pt_f4 lds_load16(uint32_t addr) { return *reinterpret_cast<lds_pt_f4*>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ void lds_store8(uint32_t addr, float a, float b) { *reinterpret_cast<lds_pt_f2*>(static_cast<uintptr_t>(addr)) = pt_f2{a, b}; }
__device__ __forceinline__ pt_f2 lds_load8(uint32_t addr) { return *reinterpret_cast<lds_pt_f2*>(static_cast<uintptr_t>(addr)); }

// SPLIT (scenes read from global memory with at least one light): the shadow walks do NOT run
// here. Every light sample is written as a request (direction, the factors of its
// contribution, the distance test's operands) and the vertex's `color +=` is deferred to
// k_shadow, which walks at k_extend's occupancy instead of this kernel's (measured: the walks
// ran 2x slower per ray inside the 128..168-register shading kernel). `shaded` tells the
// caller to queue the slot for k_shadow. The deferred arithmetic is the same expressions in
// the same order, so results stay bit-identical to the fused form.
// DEFER (k_paths_merged, at most two light samples per vertex): like SPLIT, but the requests stay on chip (`pend`, `stash`), the
// one-triangle shortcuts of all-opaque scenes are taken here, and a finished path keeps its colour and throughput in `p` until
// resolve_pending has added the vertex's direct light (it then ends the sample and resets them).
template <bool IN_LDS, bool ALL_OPAQUE, bool SPLIT, bool PUNCTUAL = true, bool TEXTURED = true, bool DEFER = false>
__device__ __forceinline__ bool shade_vertex(const pt_dev_scene& sc, const ShadeData& sd, const pt_dev_frame& f, const pt_dev_state& st,
                                             const Tracer& tr, uint32_t n_lights, vec3 bg_linear, uint32_t slot,
                                             const Hit& hit, vec3 ray_dir, PathRegs& p, vec3& next_o, vec3& next_d,
                                             Tally& tally, Tally& walks, bool& shaded, Pending* pend = nullptr, const Stash* stash = nullptr) {
  vec3 indirect = p.indirect, color = p.color;
  uint32_t rng = p.rng, light_steps = p.light_steps, transp_steps = p.transp_steps, sample = p.sample;
  bool alive = false;
  shaded = false;
  float alpha_factor = 1.0f;   // SPLIT: what `direct` is multiplied by before `* indirect` (main.comp:545 / :552)
  vec3 deferred_m = splat(0.0f), deferred_emission = splat(0.0f), deferred_origin = splat(0.0f);
  const vec3 color_in = color;
  bool path_done = false;  // the sample ends at this vertex

  if (hit.tri == kNoHit) {  // main.comp:517 -> :580
    path_done = true;
  } else {
    const vec3 wuv = mk(1.0f - hit.u - hit.v, hit.u, hit.v);
    vec3 pos, nrm;
    float tu, tv;
    uint32_t mat_idx;
    interpolate_pos_uv<IN_LDS>(sd, hit.tri, wuv, pos, tu, tv, mat_idx, &nrm, tally);
    if (dot(nrm, ray_dir) > 0.0f) nrm = nrm * -1.0f;  // main.comp:524

    // get_material_data, main.comp:242-259
    Material m = load_material<IN_LDS>(sd.materials, mat_idx);
    if (!TEXTURED) m.base_tex = m.mr_tex = m.emissive_tex = m.occlusion_tex = -1;  // no material names a texture: the sampling code folds away
    const float4 bc = sample_texture(sc, m.base_tex, tu, tv, tally);
    const vec3 base_color = mk(m.base_color.x, m.base_color.y, m.base_color.z) * decode_srgb(m.base_tex, bc);
    const float alpha = m.base_color.w * bc.w;
    const float4 mr = sample_texture(sc, m.mr_tex, tu, tv, tally);
    const float metal = m.metallic * mr.z;
    const float roughness = m.roughness * mr.y;
    const float4 em = sample_texture(sc, m.emissive_tex, tu, tv, tally);
    const vec3 emission = m.emissive * decode_srgb(m.emissive_tex, em);
    const float4 oc = sample_texture(sc, m.occlusion_tex, tu, tv, tally);
    const float occlusion = 1.0f + m.occlusion_strength * (oc.x - 1.0f);
    const float a = roughness * roughness;
    const float a2 = a * a;
    ++tally.verts;

    const vec3 view = -ray_dir;
    const vec3 shadow_origin = pos + kNormalDelta * nrm;

    // sample_punctual_light (main.comp:409-460) and sample_area_light (:464-501) as ONE loop
    // over the vertex's light samples — directional lights, point lights, then the area
    // sample, i.e. the reference's order of RNG draws and of the `+=` into `punctual` — so
    // that the kernel holds a single copy of the shadow walk (three inlined copies were
    // 16 k instructions and 168 VGPRs).
    vec3 punctual = splat(0.0f);
    vec3 area_light = splat(0.0f);
    const uint32_t n_dir = sc.directional_light_count, n_punctual = n_dir + sc.point_light_count;
    const uint32_t n_samples = PUNCTUAL ? n_lights : (n_lights != 0u ? 1u : 0u);  // !PUNCTUAL: at most one trip, no back edge
    for (uint32_t j = 0; j < n_samples; ++j) {
      vec3 l, sample_pos = splat(0.0f), sample_dir = splat(0.0f);
      float4 l0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float dist2 = 0.0f;
      uint32_t etri = 0;
      vec3 ewuv = splat(0.0f);
      float4 q0 = l0, q1 = l0, q2 = l0;
      // PUNCTUAL = false: the scene has no directional or point light, every sample is the area sample
      uint32_t kind = !PUNCTUAL ? 2u : (j < n_dir ? 0u : (j < n_punctual ? 1u : 2u));
      // (DEFER: the kind is wave-uniform; pinned to a scalar register so that every test of it below is a scalar compare. Left alone,
      // hipcc formed `j < n_punctual` as a LANE MASK (v_cndmask + v_cmp under the trip's exec mask) and re-tested that mask further
      // down the trip — the instruction pair scripts/check_isa.py exists to keep out of the kernels)
      if (DEFER && PUNCTUAL) kind = __builtin_amdgcn_readfirstlane(kind);
      if (kind == 0u) {
        l0 = fetch16<IN_LDS>(sd.dlights, 2 * j);
        l = -xyz(fetch16<IN_LDS>(sd.dlights, 2 * j + 1));
      } else if (kind == 1u) {
        const uint32_t i = j - n_dir;
        l0 = fetch16<IN_LDS>(sd.plights, 2 * i);
        const vec3 lpos = xyz(fetch16<IN_LDS>(sd.plights, 2 * i + 1));
        sample_dir = uniform_hemisphere_sampling(normalize(pos - lpos), rng);
        sample_pos = lpos + sample_dir * kPointLightRadius;
        l = sample_pos - pos;
        dist2 = dot(l, l);
        l = normalize(l);
      } else {
        const uint32_t count = f.pc.emissive_triangle_count;
        uint32_t pick = static_cast<uint32_t>(rand_float(rng) * static_cast<float>(count));
        if (pick > count - 1) pick = count - 1;  // Q3: rand_float may be exactly 1.0
        etri = fetch32<IN_LDS>(sd.emissive, pick);
        float su = rand_float(rng), sv = rand_float(rng);  // sample_triangle, main.comp:158-165
        if (su + sv > 1.0f) {
          su = 1.0f - su;
          sv = 1.0f - sv;
        }
        ewuv = mk(1.0f - su - sv, su, sv);
        // get_interpolated_data of the emitter; its normal, uv and material are only needed
        // if the walk lets the sample through, so only the position is formed here
        ++tally.hits;
        fetch16x3<IN_LDS, PT_PIN_EMIT != 0>(sd.tri_pos, 3 * etri, q0, q1, q2);
        sample_pos = mat3_mul(xyz(q0), xyz(q1), xyz(q2), ewuv);
        l = sample_pos - pos;
        dist2 = dot(l, l);
        l = normalize(l);
      }

      if (DEFER && ALL_OPAQUE) {  // the one-triangle shortcuts (below, in the fused form's words): a sample they decide is occluded and adds nothing
        Hit sh{kInfinity, 0.0f, 0.0f, kNoHit};
        if (kind == 2u) {
          ++walks.tests;
          ray_triangle<IN_LDS>(tr, shadow_origin, l, etri, sh);
        }
        if (sh.tri == kNoHit && !(kind == 2u && etri == hit.tri)) {
          ++walks.tests;
          ray_triangle<IN_LDS>(tr, shadow_origin, l, hit.tri, sh);
        }
        if (sh.tri != kNoHit) {
          ++walks.rays;
          ++walks.paths;
          continue;
        }
      }
      if (SPLIT || DEFER) {
        vec3 a_factor;       // the radiance / emission factor that transparency multiplies
        float n_dot_l, pdf = 1.0f, limit = -kInfinity * 2.0f, correction = 0.0f;  // -inf: `x < limit` never holds (no distance test for directional lights)
        uint32_t tap_if_visible = 0;
        if (kind == 0u) {
          n_dot_l = gmax(dot(nrm, l), 0.0f);
          a_factor = xyz(l0) * l0.w;
        } else {
          correction = sdiv(kNormalDelta, dot(l, nrm));
          limit = length(pos - sample_pos);
          n_dot_l = __builtin_fabsf(dot(nrm, l));
          if (kind == 1u) {
            a_factor = xyz(l0) * l0.w / kLightArea / kFourPi;
            const float cos_theta0 = dot(sample_dir, -l);
            pdf = sdiv(dist2, cos_theta0 * kLightArea);
          } else {
            const TriAttr at = load_attr<IN_LDS>(sd.tri_attr, etri);
            vec3 tri_n = normalize(mat3_mul(at.n0, at.n1, at.n2, ewuv));
            const float eu = at.u0 * ewuv.x + at.u1 * ewuv.y + at.u2 * ewuv.z;
            const float ev = at.v0 * ewuv.x + at.v1 * ewuv.y + at.v2 * ewuv.z;
            if (dot(l, tri_n) > 0.0f) tri_n = tri_n * -1.0f;
            const float area = sdiv(length(cross(xyz(q1) - xyz(q0), xyz(q2) - xyz(q0))), 2.0f);
            const float cos_theta0 = dot(tri_n, -l);
            pdf = sdiv(sdiv(dist2, cos_theta0 * area), static_cast<float>(f.pc.emissive_triangle_count));
            Material emat = load_material<IN_LDS>(sd.materials, at.mat);
            if (!TEXTURED) emat.emissive_tex = -1;
            Tally uncounted;  // the reference fetches the emitter's texture only for visible samples: k_shadow counts the tap
            const float4 et = sample_texture(sc, emat.emissive_tex, eu, ev, uncounted);
            tap_if_visible = emat.emissive_tex != -1 ? 1u : 0u;
            a_factor = emat.emissive * decode_srgb(emat.emissive_tex, et);
          }
        }
        const vec3 brdf = microfacet_brdf(nrm, view, l, metal, a2, base_color);
        if (DEFER) {  // slot j (the host sends scenes with more than two light samples per vertex to k_paths)
          if (j == 0u) pend->l0 = l;
          else pend->l1 = l;
          pend->flags |= (PEND_S0 << j) | (kind == 2u ? PEND_AREA0 << j : 0u) | (tap_if_visible ? PEND_TAP0 << j : 0u);
          lds_store16(stash->quad + 2u * j * stash->quad_stride, a_factor.x, a_factor.y, a_factor.z, correction);
          lds_store16(stash->quad + (2u * j + 1u) * stash->quad_stride, brdf.x, brdf.y, brdf.z, n_dot_l);
          lds_store8(stash->pair + j * stash->pair_stride, pdf, limit);
          continue;
        }
        float4* r = st.req + (static_cast<size_t>(j) * 4u) * st.slot_stride + slot;
        r[0] = make_float4(l.x, l.y, l.z, limit);
        r[st.slot_stride] = make_float4(a_factor.x, a_factor.y, a_factor.z, correction);
        r[2u * static_cast<size_t>(st.slot_stride)] = make_float4(brdf.x, brdf.y, brdf.z, n_dot_l);
        r[3u * static_cast<size_t>(st.slot_stride)] = make_float4(pdf, ubits(etri), ubits(tap_if_visible), 0.0f);
        continue;
      }

      Hit sh{kInfinity, 0.0f, 0.0f, kNoHit};
      if (ALL_OPAQUE && kind == 2u) {
        // Any accepted triangle along the ray decides an all-opaque shadow query (Q5), and an
        // area sample aims at a known triangle: test that one first. Most samples seen from
        // its front are accepted by it and never walk the tree (the oracle's checker mode
        // mirrors the shortcut, so the visit/test counters still compare equal).
        ++walks.tests;
        ray_triangle<IN_LDS>(tr, shadow_origin, l, etri, sh);
        if (sh.tri != kNoHit) {
          ++walks.rays;
          ++walks.paths;  // in the shadow slice this word counts the rays decided by the shortcut (pt_stats.rays_shadow_shortcut)
          continue;
        }
      }
      if (ALL_OPAQUE && !IN_LDS && !(kind == 2u && etri == hit.tri)) {
        // ... and so does the triangle the shading point lies on: a light on the far side of the surface sends the ray, which
        // starts NORMAL_DELTA above that triangle, straight back through it, and the walk would need a full descent of the tree
        // to find that out (the 5 M-triangle rocks: 51 % of the shadow rays, -12.8 % node visits; ORC_FLAG_SELF_FIRST mirrors it).
        // Scenes read from global memory only (an LDS-resident scene's walk is a handful of ds_reads; cfg2 lost 3 % to the extra code).
        ++walks.tests;
        ray_triangle<IN_LDS>(tr, shadow_origin, l, hit.tri, sh);
        if (sh.tri != kNoHit) {
          ++walks.rays;
          ++walks.paths;
          continue;
        }
      }
      const float transparency = shadow_walk<IN_LDS, ALL_OPAQUE, TEXTURED>(sc, sd, tr, shadow_origin, l, sh, walks);
      if (transparency == 0.0f) continue;
      if (kind != 0u) {
        const float correction = sdiv(kNormalDelta, dot(l, nrm));
        if (sh.tri != kNoHit && sh.t + correction < length(pos - sample_pos)) continue;
      }
      if (kind == 0u) {
        const float n_dot_l = gmax(dot(nrm, l), 0.0f);
        const vec3 radiance = xyz(l0) * l0.w * transparency;
        punctual = punctual + radiance * microfacet_brdf(nrm, view, l, metal, a2, base_color) * n_dot_l;
      } else if (kind == 1u) {
        const vec3 radiance = xyz(l0) * l0.w / kLightArea / kFourPi * transparency;
        const float cos_theta0 = dot(sample_dir, -l);
        const float pdf = sdiv(dist2, cos_theta0 * kLightArea);
        const float n_dot_l = __builtin_fabsf(dot(nrm, l));
        punctual = punctual + radiance * microfacet_brdf(nrm, view, l, metal, a2, base_color) * n_dot_l / pdf;
      } else {
        const TriAttr at = load_attr<IN_LDS>(sd.tri_attr, etri);
        vec3 tri_n = normalize(mat3_mul(at.n0, at.n1, at.n2, ewuv));
        const float eu = at.u0 * ewuv.x + at.u1 * ewuv.y + at.u2 * ewuv.z;
        const float ev = at.v0 * ewuv.x + at.v1 * ewuv.y + at.v2 * ewuv.z;
        if (dot(l, tri_n) > 0.0f) tri_n = tri_n * -1.0f;
        const float area = sdiv(length(cross(xyz(q1) - xyz(q0), xyz(q2) - xyz(q0))), 2.0f);
        const float cos_theta0 = dot(tri_n, -l);
        const float pdf = sdiv(sdiv(dist2, cos_theta0 * area), static_cast<float>(f.pc.emissive_triangle_count));
        const float n_dot_l = __builtin_fabsf(dot(nrm, l));
        Material emat = load_material<IN_LDS>(sd.materials, at.mat);
            if (!TEXTURED) emat.emissive_tex = -1;
        const float4 et = sample_texture(sc, emat.emissive_tex, eu, ev, tally);
        const vec3 e_emission = emat.emissive * decode_srgb(emat.emissive_tex, et) * transparency;
        area_light = e_emission * microfacet_brdf(nrm, view, l, metal, a2, base_color) * n_dot_l / pdf;
      }
    }

    vec3 direct = emission;        // main.comp:536
    direct = direct + punctual;    // :537
    direct = direct + area_light;  // :538
    indirect = indirect * occlusion;  // :540
    if (SPLIT) {
      shaded = true;
      deferred_m = indirect;
      deferred_emission = emission;
      deferred_origin = shadow_origin;
    }
    if (DEFER) {
      shaded = true;
      pend->origin = shadow_origin;
      pend->emission = emission;
      pend->m = indirect;
      pend->af = 1.0f;
      pend->flags |= PEND_SHADED;
    }

    // (a provably all-opaque scene has alpha == 1.0f at every vertex: the pass-through arm is dead)
    if (!ALL_OPAQUE && alpha < 1.0f && transp_steps < kMaxTransparencySteps) {  // main.comp:542-547
      ++transp_steps;
      if (SPLIT) alpha_factor = alpha;
      else if (DEFER) pend->af = alpha;
      else color = color + direct * alpha * indirect;
      indirect = indirect * (1.0f - alpha);
      next_o = pos + kNormalDelta * ray_dir;
      next_d = ray_dir;
    } else {  // main.comp:549-577
      ++light_steps;
      if (!SPLIT && !DEFER) color = color + direct * indirect;
      const Basis basis = orthogonal_matrix(nrm);  // shared by both samplers (:383, :399)
      float diffuse_pdf, microfacet_pdf;
      const vec3 diffuse_dir = diffuse_sample(nrm, basis, rng, diffuse_pdf);
      const vec3 microfacet_dir = microfacet_sample(basis, view, a2, rng, microfacet_pdf);
      vec3 dir;
      float c;
      if (rand_float(rng) < metal) {
        dir = microfacet_dir;
        c = metal;
      } else {
        dir = diffuse_dir;
        c = 1.0f - metal;
      }
      const float pdf = c * (diffuse_pdf + microfacet_pdf);
      const float n_dot_l = gmax(dot(nrm, dir), 0.0f);
      indirect = indirect * (microfacet_brdf(nrm, view, dir, metal, a2, base_color) * n_dot_l / pdf);
      next_o = shadow_origin;
      next_d = dir;
    }
    path_done = light_steps >= kMaxLightSteps;  // main.comp:512
  }

  if (SPLIT && shaded) {  // k_shadow finishes this vertex: color_in + direct * alpha_factor * deferred_m, then the sample if it ends here
    float4* h = st.pend + slot;
    const size_t ss = st.slot_stride;
    h[0] = make_float4(deferred_emission.x, deferred_emission.y, deferred_emission.z, alpha_factor);
    h[ss] = make_float4(deferred_m.x, deferred_m.y, deferred_m.z, ubits(path_done ? 1u : 0u));
    h[2 * ss] = make_float4(deferred_origin.x, deferred_origin.y, deferred_origin.z, 0.0f);
    h[3 * ss] = make_float4(color_in.x, color_in.y, color_in.z, 0.0f);
    if (path_done) h[4 * ss] = make_float4(indirect.x, indirect.y, indirect.z, 0.0f);
  }

  if (path_done) {  // end of trace() + next stratum of this pixel, or retire the slot (main():597-612)
    if (DEFER && shaded) pend->flags |= PEND_ENDED;  // resolve_pending ends the sample once the vertex's direct light is in
    if (!((SPLIT || DEFER) && shaded)) {
      float4 acc = st.accum[slot];
      finish_sample(color, indirect, bg_linear, acc, tally);
      st.accum[slot] = acc;
    }
    ++sample;
    if (sample < f.sample_end) {
      uint32_t x, y;
      slot_to_pixel(f, slot, x, y);
      camera_ray(f, x, y, sample, rng, next_o, next_d);
      ++tally.paths;
      alive = true;
    }
    if (!(DEFER && shaded)) {  // (DEFER: the finished path's colour and throughput stay in p until resolve_pending has used them)
      indirect = splat(1.0f);
      color = splat(0.0f);
    }
    light_steps = 0;
    transp_steps = 0;
  } else {
    alive = true;
  }

  p.indirect = indirect;
  p.color = color;
  p.rng = rng;
  p.light_steps = light_steps;
  p.transp_steps = transp_steps;
  p.sample = sample;
  return alive;
}

// Waves per SIMD the register allocator must leave room for (measured): LDS-resident
// scenes are VALU-issue bound and run best with the 168 registers of 3 waves; scenes read from
// global memory are latency bound and gain more from a 4th wave than they lose to spills.
#ifndef PT_SHADE_MIN_WAVES_LDS
#define PT_SHADE_MIN_WAVES_LDS 3
#endif
#ifndef PT_SHADE_MIN_WAVES_GLOBAL
#define PT_SHADE_MIN_WAVES_GLOBAL 4
#endif
#define PT_SHADE_MIN_WAVES(in_lds) ((in_lds) ? PT_SHADE_MIN_WAVES_LDS : PT_SHADE_MIN_WAVES_GLOBAL)
// k_paths also runs the closest-hit traversals: swept 2..8 on cfg2, best at 4 (128 registers; 5 and more spill)
#ifndef PT_PATHS_MIN_WAVES
#define PT_PATHS_MIN_WAVES 4
#endif
#ifndef PT_PATHS_MIN_WAVES_PLAIN
#define PT_PATHS_MIN_WAVES_PLAIN PT_PATHS_MIN_WAVES
#endif
#ifndef PT_PATHS_MIN_WAVES_GLOBAL
#define PT_PATHS_MIN_WAVES_GLOBAL 5  // scenes beyond an XCD's L2: swept 3..6 on cfg3/cfg4-class scenes, best at 5
#endif
#ifndef PT_FAST_ARITH
namespace PT_KNS {  // kernels live in the build's own namespace: the same template compiled twice must not share a symbol
// shade_vertex for every extended ray, then compaction of the survivors into the next queue.
// Persistent workgroups striding the queue, so the LDS scene copy is staged once per block.
template <bool IN_LDS, bool ALL_OPAQUE, bool SPLIT>
__global__ __launch_bounds__(256, PT_SHADE_MIN_WAVES(IN_LDS)) void k_shade(pt_dev_scene sc, pt_dev_frame f, pt_dev_state st, uint32_t parity,
                                                                             uint32_t compact) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = st.counters[PT_CNT_QUEUE0 + parity];
  Tally tally;   // shading work
  Tally walks;   // the shadow walks' traversal work (reported as the `shadow` slice)
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (f.pc.emissive_triangle_count ? 1u : 0u);
  Tracer tr{};
  ShadeData sd{sc.tri_pos, sc.tri_attr, sc.materials, sc.directional_lights, sc.point_lights, sc.emissive_triangles};
  if (blockIdx.x * blockDim.x < n) {  // block-uniform
    if (IN_LDS || (!SPLIT && n_lights)) tr = make_tracer<IN_LDS>(sc, lds_dyn);
    sd = make_shade_data<IN_LDS>(sc, tr, lds_dyn);
  }
  const vec3 bg_linear = mk(f.bg_linear[0], f.bg_linear[1], f.bg_linear[2]);

  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {  // block-uniform trip count
    const uint32_t gid = base + threadIdx.x;
    const bool active = gid < n;
    bool alive = false;   // slot continues with a ray next iteration
    bool shaded = false;  // SPLIT: slot has a deferred direct-light update for k_shadow
    uint32_t slot = 0;

    float4 pb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool live = false;  // not retired since the queue was last compacted
    if (active) {
      slot = st.queue[parity][gid];
      pb = st.path_b[slot];
      live = ctr_sample(fbits(pb.w)) < f.sample_end;
    }
    if (live) {
      const uint4 hw = st.hit[slot];
      const float4 pa = st.path_a[slot];
      const uint32_t ctr = fbits(pb.w);
      PathRegs p{xyz(pa), xyz(pb), fbits(pa.w), ctr_light(ctr), ctr_transp(ctr), ctr_sample(ctr)};
      if (ctr_primary(ctr)) st.primary_id[slot] = hw.x;
      const Hit hit{ubits(hw.w), ubits(hw.y), ubits(hw.z), hw.x};
      vec3 ray_dir = mk(0, 0, 0);
      if (hw.x != kNoHit) ray_dir = xyz(st.ray_d[slot]);
      vec3 next_o = mk(0, 0, 0), next_d = mk(0, 0, 0);
      alive = shade_vertex<IN_LDS, ALL_OPAQUE, SPLIT>(sc, sd, f, st, tr, n_lights, bg_linear, slot, hit, ray_dir, p, next_o, next_d, tally,
                                                      walks, shaded);
      if (alive) {
        st.ray_o[slot] = make_float4(next_o.x, next_o.y, next_o.z, 0.0f);
        st.ray_d[slot] = make_float4(next_d.x, next_d.y, next_d.z, 0.0f);
      } else {
        st.ray_d[slot] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);  // retired: k_extend skips the slot until the next compaction drops it
      }
      st.path_a[slot] = make_float4(p.indirect.x, p.indirect.y, p.indirect.z, ubits(p.rng));
      st.path_b[slot] = make_float4(p.color.x, p.color.y, p.color.z, ubits(ctr_pack(p.light_steps, p.transp_steps, false, p.sample)));
    }

    // Queue compaction (the whole block converges here). A slot leaves the queue only when its
    // pixel has run all its samples, so the queue changes slowly: it is rebuilt on the
    // iterations the host marks (`compact`, block-uniform) and re-used as it stands in between
    // — no barrier, no atomic, no queue write on those iterations; retired slots idle.
    if (compact) {
      const AppendSlots at = block_append2(alive, &st.counters[PT_CNT_QUEUE0 + (parity ^ 1u)], SPLIT && shaded,
                                           SPLIT ? &st.counters[PT_CNT_SHADOW] : nullptr);
      if (alive) st.queue[parity ^ 1u][at.a] = slot;
      if (SPLIT && shaded) st.queue_shadow[at.b] = slot;
    }
  }
  flush_tally(tally, st, PT_K_SHADE);
  if (!SPLIT) flush_tally(walks, st, PT_K_SHADOW);
}

// The deferred half of a SPLIT k_shade: the shadow walks of every light sample of every
// vertex shaded this iteration (ray_scene_intersect_with_transparency, main.comp:306-332),
// the direct-lighting accumulate (main.comp:536-552) and, where the vertex ended its path,
// the end of trace() and the NaN filter (main.comp:580-582, 607-612). A lean traversal
// kernel like k_extend; persistent workgroups striding the shadow queue.
// occupancy bound measured per variant: the any-hit walk fits 4 waves without help and
// ran best there; the closest-hit walk with its material/texture fetch gains from being held to 6
#ifndef PT_SHADOW_MIN_WAVES_OPAQUE
#define PT_SHADOW_MIN_WAVES_OPAQUE 4
#endif
#ifndef PT_SHADOW_MIN_WAVES_ALPHA
#define PT_SHADOW_MIN_WAVES_ALPHA 6
#endif
template <bool ALL_OPAQUE>
__global__ __launch_bounds__(256, ALL_OPAQUE ? PT_SHADOW_MIN_WAVES_OPAQUE : PT_SHADOW_MIN_WAVES_ALPHA) void k_shadow(pt_dev_scene sc, pt_dev_frame f, pt_dev_state st) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = st.counters[PT_CNT_SHADOW];
  Tally walks;
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (f.pc.emissive_triangle_count ? 1u : 0u);
  const vec3 bg_linear = mk(f.bg_linear[0], f.bg_linear[1], f.bg_linear[2]);
  const size_t ss = st.slot_stride;
  // (n_lights != 0 tested HERE, with the scalar unit: the host only launches this kernel for scenes with lights, and left to the light
  // loop's own guard the compiler turned the wave-uniform comparison into a lane mask in the prologue — scripts/check_isa.py's pattern)
  if (blockIdx.x * blockDim.x < n && n_lights != 0u) {  // block-uniform
    const Tracer tr = make_tracer<false>(sc, lds_dyn);
    const ShadeData sd = make_shade_data<false>(sc, tr, lds_dyn);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const uint32_t slot = st.queue_shadow[i];
      const float4* h = st.pend + slot;
      const float4 h0 = h[0], h1 = h[ss], h2 = h[2 * ss], h3 = h[3 * ss];
      const vec3 origin = xyz(h2);
      const uint32_t self_tri = ALL_OPAQUE ? st.hit[slot].x : kNoHit;  // the triangle the vertex lies on (k_extend's result, untouched until the next iteration)
      vec3 punctual = splat(0.0f), area_light = splat(0.0f);
      for (uint32_t j = 0; j < n_lights; ++j) {
        const float4* r = st.req + (static_cast<size_t>(j) * 4u) * ss + slot;
        const float4 r0 = r[0], r1 = r[ss], r2 = r[2 * ss], r3 = r[3 * ss];
        const vec3 l = xyz(r0);
        const bool is_area = f.pc.emissive_triangle_count && j + 1 == n_lights;
        Hit sh{kInfinity, 0.0f, 0.0f, kNoHit};
        if (ALL_OPAQUE && is_area) {  // the emitter-first shortcut of the fused form
          ++walks.tests;
          ray_triangle<false>(tr, origin, l, fbits(r3.y), sh);
          if (sh.tri != kNoHit) {
            ++walks.rays;
            ++walks.paths;  // shortcut count (see shade_vertex)
            continue;
          }
        }
        if (ALL_OPAQUE && !(is_area && fbits(r3.y) == self_tri)) {  // the own-triangle-first shortcut of the fused form
          ++walks.tests;
          ray_triangle<false>(tr, origin, l, self_tri, sh);
          if (sh.tri != kNoHit) {
            ++walks.rays;
            ++walks.paths;
            continue;
          }
        }
        const float transparency = shadow_walk<false, ALL_OPAQUE>(sc, sd, tr, origin, l, sh, walks);
        if (transparency == 0.0f) continue;
        if (sh.tri != kNoHit && sh.t + r1.w < r0.w) continue;  // main.comp:447 / :487 (never holds for a directional light)
        walks.taps += fbits(r3.z);
        const vec3 c = xyz(r1) * transparency * xyz(r2) * r2.w / r3.x;
        if (is_area) area_light = c;
        else punctual = punctual + c;
      }
      vec3 direct = xyz(h0);          // main.comp:536
      direct = direct + punctual;     // :537
      direct = direct + area_light;   // :538
      const vec3 color = xyz(h3) + direct * h0.w * xyz(h1);  // :545 / :552 (h0.w == 1 for the opaque branch)
      if (fbits(h1.w) & 1u) {         // the path ended at this vertex: main.comp:580-582, 607-612
        float4 acc = st.accum[slot];
        finish_sample(color, xyz(h[4 * ss]), bg_linear, acc, walks);
        st.accum[slot] = acc;
      } else {
        float4 pb = st.path_b[slot];
        pb.x = color.x;
        pb.y = color.y;
        pb.z = color.z;
        st.path_b[slot] = pb;
      }
    }
  }
  flush_tally(walks, st, PT_K_SHADOW);
}

}  // namespace PT_KNS
#endif  // !PT_FAST_ARITH

// ---------------------------------------------------------------- the merged walk (k_paths_merged)
// A lane's traversal jobs of one phase, walked one after the other in ONE loop: the pending light samples of the vertex it shaded
// last (slots 0 and 1), then the closest hit of its next ray. All-opaque scenes: the shadow jobs are any-hit walks (first accepted
// triangle decides; children entered longer-span-first), the closest-hit job keeps the reference's order (Q14). Other scenes: every
// job is a closest-hit walk, and a shadow job is the FIRST step of ray_scene_intersect_with_transparency (main.comp:306-332): a miss
// lets the sample through, a provably opaque hit (tri_pos tag) blocks it, and only a hit on a surface that may be transparent goes on
// with the reference's step-by-step walk (shadow_walk<RESUME>, lanes in lock step: rare — cloth, glass).
// A lane that finishes a job parks. Parked lanes are served together — results booked, next job started (three IEEE divides for
// 1 / d) — once kMergedPark<> of them wait or nobody is walking: serving each lane the moment it parks would run that block up to 192
// times per wave and phase for a lane or two at a time. Measured (same box, 1080p, rocks 64 spp / atrium 36 spp, Mrays/s; k_paths
// 1511 / 1878): park 1: 1598 / 1857, 2: 1569 / 1874, 6: 1555 / 1924, 8: 1549 / 1933, 12: 1530 / 1928, 24: 1455 / 1861 (11 LDS rows).
#ifndef PT_MERGED_PARK
#define PT_MERGED_PARK 8
#endif
// Waves per SIMD the register allocator leaves room for in k_paths_merged: 5 (96 registers), as k_paths on these scenes. The first
// version ran 3 x slower there: the four quads of a pair record, read as one 64-byte vector, lived in ONE 16-register tuple that was
// spilled and reloaded whole on every inner-node trip to keep its last quad across the slab test (read in the ISA,
// scripts/isa_fn.py); fetched as four pinned 128-bit values (fetch16x4) the loop holds 0-3 one-word spills. At 4 waves (128
// registers, no spill at all) it only equals k_paths at 5.
#ifndef PT_MERGED_MIN_WAVES
#define PT_MERGED_MIN_WAVES 5
#endif
// 1: path state, deferred update and job rays parked in private memory around the walk and read back from there (tried against the
// spills above before their cause was found; scratch latency at every job start costs more than the registers: -8 % / -17 %)
#ifndef PT_MERGED_COLD
#define PT_MERGED_COLD 0
#endif
// All-opaque scenes (cfg4): shadow jobs are any-hit walks, short, and their serving block is light (no first hit to keep): served at
// once, without the vote. Threshold 8 / 4 / 2 / 1: rocks 1615 / 1635 / 1637 / 1668, soup 1639 / 1674 / 1687 / 1707 Mrays/s (9 LDS rows).
#ifndef PT_MERGED_PARK_OPAQUE
#define PT_MERGED_PARK_OPAQUE 1
#endif
template <bool ALL_OPAQUE>
constexpr uint32_t kMergedPark = ALL_OPAQUE ? PT_MERGED_PARK_OPAQUE : PT_MERGED_PARK;
// Experiment: when a far child is deferred, touch one word of what its later visit will fetch (its pair record, or its first
// triangle), so that the line is on its way when the entry is popped. 1: every deferred child, 2: inner children only.
#ifndef PT_TOUCH_FAR
#define PT_TOUCH_FAR 0
#endif

//
// `cold` is a private (scratch) array: it holds the first hit of a light sample that may be transparent until the walk is over
// (COLD_HIT), and with PT_MERGED_COLD the path state, the deferred colour update and the jobs' rays around the walk. Its words:
enum : uint32_t {
  COLD_INDIRECT = 0, COLD_COLOR = 3, COLD_RNG = 6, COLD_CTR = 7, COLD_EMISSION = 8, COLD_M = 11, COLD_AF = 14,
  COLD_ORIGIN = 15, COLD_L0 = 18, COLD_L1 = 21, COLD_RAY_O = 24, COLD_RAY_D = 27, COLD_HIT = 30 /* 2 x {t, u, v, tri} */, COLD_WORDS = 38
};
__device__ __forceinline__ void cold_put(volatile float* cold, uint32_t at, vec3 v) {
  cold[at] = v.x;
  cold[at + 1] = v.y;
  cold[at + 2] = v.z;
}
__device__ __forceinline__ vec3 cold_get(volatile float* cold, uint32_t at) {
  const float x = cold[at], y = cold[at + 1], z = cold[at + 2];
  return mk(x, y, z);
}

// ---- PT_BVH=wide4 (opt-in; NOT bit-exact: validated by north_star's tolerance like PT_ARITH=fast) ---------------------------------
// The walk is bound by the number of divergent 16-byte lane loads, four per inner-node visit. A wide record puts up
// to FOUR children into the same 64 bytes (built at upload by collapsing the reference's binary tree, pt_context.hip): a common
// origin and per-axis power-of-two scale, 8-bit child boxes rounded outwards, four child entries (ref << 7 | cnt) — half the visits
// per ray, so half the loads, and a third of the node bytes. The boxes are supersets of the reference's and the children are
// entered nearest first, so the closest hit is the reference's except where two candidates tie exactly or a ray grazes a box the
// reference's slab test misses by a rounding; any-hit walks take the children in record order.
// Record: {origin.xyz, ex | ey << 8 | ez << 16 (biased exponents)} {entry 0..3} {lo.x[4], lo.y[4], lo.z[4], hi.x[4]} {hi.y[4], hi.z[4], -, -}.
constexpr uint32_t kDeepLevelsWide = 128;  // a wide visit defers up to three children
struct WideHits {
  float key[4];      // entry distance of child c if the ray enters its box before the closest hit so far, else +inf
  uint32_t entry[4];
};
__device__ __forceinline__ float ubyte_f(uint32_t w, uint32_t c) { return static_cast<float>((w >> (8u * c)) & 255u); }  // v_cvt_f32_ubyteN
__device__ __forceinline__ WideHits wide_visit(vec3 o, vec3 inv, const float4& q0, const float4& q1, const float4& q2, const float4& q3, float t_best) {
  const uint32_t ex = fbits(q0.w);
  const vec3 scale = mk(ubits((ex & 255u) << 23), ubits(((ex >> 8) & 255u) << 23), ubits(((ex >> 16) & 255u) << 23));
  const vec3 a = mk(scale.x * inv.x, scale.y * inv.y, scale.z * inv.z);
  const vec3 b = mk((q0.x - o.x) * inv.x, (q0.y - o.y) * inv.y, (q0.z - o.z) * inv.z);
  const uint32_t lox = fbits(q2.x), loy = fbits(q2.y), loz = fbits(q2.z), hix = fbits(q2.w), hiy = fbits(q3.x), hiz = fbits(q3.y);
  const uint32_t e[4] = {fbits(q1.x), fbits(q1.y), fbits(q1.z), fbits(q1.w)};
  WideHits r;
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) {
    const float x0 = __builtin_fmaf(ubyte_f(lox, c), a.x, b.x), x1 = __builtin_fmaf(ubyte_f(hix, c), a.x, b.x);
    const float y0 = __builtin_fmaf(ubyte_f(loy, c), a.y, b.y), y1 = __builtin_fmaf(ubyte_f(hiy, c), a.y, b.y);
    const float z0 = __builtin_fmaf(ubyte_f(loz, c), a.z, b.z), z1 = __builtin_fmaf(ubyte_f(hiz, c), a.z, b.z);
    const float near = gmax(gmax(gmin(x0, x1), gmin(y0, y1)), gmin(z0, z1));
    const float far = gmin(gmin(gmax(x0, x1), gmax(y0, y1)), gmax(z0, z1));
    const bool hit = e[c] != 0xFFFFFFFFu && near <= far && !(far < 0.0f) && near < t_best;
    r.key[c] = hit ? near : kInfinity * 2.0f;
    r.entry[c] = e[c];
  }
  return r;
}
__device__ __forceinline__ void wide_cswap(WideHits& r, int i, int j) {  // key[i] <= key[j] afterwards
  const bool sw = r.key[j] < r.key[i];
  const float ki = sw ? r.key[j] : r.key[i], kj = sw ? r.key[i] : r.key[j];
  const uint32_t ei = sw ? r.entry[j] : r.entry[i], ej = sw ? r.entry[i] : r.entry[j];
  r.key[i] = ki; r.key[j] = kj; r.entry[i] = ei; r.entry[j] = ej;
}

template <bool ALL_OPAQUE, bool TEXTURED, bool WIDE = false>
__device__ __forceinline__ void merged_walk(const pt_dev_scene& sc, const ShadeData& sd, const Tracer& tr, const Stash& stash,
                                            const Pending& pend, bool has_ray, vec3 ray_o, vec3 ray_d, volatile float* cold, Hit& h,
                                            float& vis0, float& vis1, Tally& ext, Tally& walks, Tally& tally) {
  uint32_t todo = (pend.flags & (PEND_S0 | PEND_S1)) | (has_ray ? 4u : 0u);  // bit 0 / 1: light sample 0 / 1, bit 2: the next ray (always last: h is its hit)
  uint32_t cur = 0;                                                          // the job being walked (one bit), 0: none
  vec3 o = splat(0.0f), d = splat(0.0f), inv = splat(0.0f);
  uint32_t sp = tr.stack, ref = 0, cnt = 0, nd = 0;
  bool live = false;
  constexpr uint32_t kDeep = WIDE ? kDeepLevelsWide : kDeepLevels;
  uint32_t deep[kDeep];  // rows beyond the lane's LDS rows (private memory), as in traverse()
#if PT_TOUCH_FAR
  uint32_t touched = 0;  // the word a touch load returned: only ever handed to an empty asm, one trip later
#endif
  auto push = [&](uint32_t e) {
    if (sp != tr.stack_end) {
      stack_store(sp, e);
      sp += tr.stride;
    } else {
      deep[nd++] = e;
    }
  };
  auto start = [&]() {         // the lane's next job (lowest bit of todo)
    cur = todo & (0u - todo);
    todo ^= cur;
#if PT_MERGED_COLD
    o = cold_get(cold, cur == 4u ? COLD_RAY_O : COLD_ORIGIN);
    d = cold_get(cold, cur == 1u ? COLD_L0 : (cur == 2u ? COLD_L1 : COLD_RAY_D));
#else
    o = cur == 4u ? ray_o : pend.origin;
    d = cur == 1u ? pend.l0 : (cur == 2u ? pend.l1 : ray_d);
#endif
    inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    h.t = kInfinity;
    h.tri = kNoHit;
    sp = tr.stack;
    nd = 0;
    ref = tr.root_ref;  // root: never box-tested (Q13)
    cnt = tr.root_cnt;
    live = true;
    if (cur == 4u) ++ext.rays;
    else ++walks.rays;
  };
  if (todo) start();
  for (;;) {
    if (live) {  // one trip of the lane's walk: the two arms of traverse()'s global-memory loop
      const bool any_hit = ALL_OPAQUE && cur != 4u;
      bool need_pop = true;
#ifdef PT_GUARD
      if ((cnt > 0 ? (ref + cnt > tr.g_tris) : (ref >= (WIDE ? tr.g_wide : tr.g_pairs))) || nd + (WIDE ? 3u : 0u) > kDeep || sp < tr.stack || sp > tr.stack_end) {
        printf("[guard] merged walk: ref %u cnt %u nd %u sp %u job %u (block %u thread %u)\n", ref, cnt, nd, sp, cur, blockIdx.x, threadIdx.x);
        live = false;
        need_pop = false;
        h.tri = kNoHit;
      } else
#endif
      if (cnt > 0) {
        if (ALL_OPAQUE) {
          for (uint32_t i = 0; i < cnt; ++i) {
            ++ext.tests;
            ray_triangle<false>(tr, o, d, ref + i, h);
            if (any_hit && h.tri != kNoHit) {  // a shadow job of an all-opaque scene: decided
              live = false;
              need_pop = false;
              break;
            }
          }
        } else {
          ext.tests += cnt;
          for (uint32_t i = 0; i < cnt; ++i) ray_triangle<false, true>(tr, o, d, ref + i, h);
        }
      } else if constexpr (WIDE && PT_STACK_CHILD != 0) {
        ++ext.visits;
        float4 q0, q1, q2, q3;
        fetch16x4<false, true>(tr.wide, 4 * ref, q0, q1, q2, q3);
        WideHits w = wide_visit(o, inv, q0, q1, q2, q3, h.t);
        wide_cswap(w, 0, 1);  // nearest child first (sorting network), the others deferred farthest first
        wide_cswap(w, 2, 3);
        wide_cswap(w, 0, 2);
        wide_cswap(w, 1, 3);
        wide_cswap(w, 1, 2);
        if (w.key[0] < kInfinity * 2.0f) {
          if (w.key[3] < kInfinity * 2.0f) push(w.entry[3]);
          if (w.key[2] < kInfinity * 2.0f) push(w.entry[2]);
          if (w.key[1] < kInfinity * 2.0f) push(w.entry[1]);
          ref = w.entry[0] >> 7;
          cnt = w.entry[0] & 127u;
          need_pop = false;
        }
      } else {
        ++ext.visits;
        // (the four quads as four 128-bit values: read as one 64-byte vector, the register allocator kept the record as ONE 16-register
        // tuple and spilled / reloaded the whole tuple around the slab test to keep its last quad, 64 + 128 bytes per trip)
        float4 q0, q1, q2, q3;
        fetch16x4<false, true>(tr.pairs, 4 * ref, q0, q1, q2, q3);
        float dist_l, dist_r, exit_l, exit_r;
        bool hit_l, hit_r;
        ray_box_pair(o, inv, q0, q1, q2, h.t, hit_l, hit_r, dist_l, dist_r, exit_l, exit_r);
        if (hit_l || hit_r) {
          bool go_left;
          if constexpr (ALL_OPAQUE) go_left = hit_l && (!hit_r || (any_hit ? left_first<true>(dist_l, dist_r, exit_l, exit_r) : dist_l < dist_r));
          else go_left = hit_l && (!hit_r || dist_l < dist_r);
          if (hit_l && hit_r) {
#if PT_STACK_CHILD  // the far side: the child itself, or (record, side)
            const uint32_t e = ((go_left ? fbits(q3.z) : fbits(q3.x)) << 7) | (go_left ? fbits(q3.w) : fbits(q3.y));
#else
            const uint32_t e = (ref << 1) | (go_left ? 1u : 0u);
#endif
            push(e);
#if PT_TOUCH_FAR && PT_STACK_CHILD
            asm volatile("" ::"v"(touched));  // (the previous touch has arrived: this trip's record was fetched after it)
            if (PT_TOUCH_FAR == 1 || (e & 127u) == 0u) {
              const float4* at = (e & 127u) ? tr.tri_pos + 3u * static_cast<size_t>(e >> 7) : tr.pairs + 4u * static_cast<size_t>(e >> 7);
              touched = *reinterpret_cast<const uint32_t*>(at);  // (a plain load: volatile would make it system-coherent, past the caches)
            }
#endif
          }
          ref = go_left ? fbits(q3.x) : fbits(q3.z);
          cnt = go_left ? fbits(q3.y) : fbits(q3.w);
          need_pop = false;
        }
      }
      if (need_pop) {
        uint32_t e = 0;
        if (nd != 0) {
          e = deep[--nd];
        } else if (sp == tr.stack) {
          live = false;  // the walk is over
        } else {
          sp -= tr.stride;
          e = stack_load(sp);
        }
        if (live) {
#if PT_STACK_CHILD
          ref = e >> 7;
          cnt = e & 127u;
#else
          const uint2 c = fetch_child<false>(tr.pairs, e >> 1, e & 1u);
          ref = c.x;
          cnt = c.y;
#endif
        }
      }
    }
    const bool parked = !live && cur != 0u;
    const uint64_t parked_mask = __ballot(parked);  // (a threshold of 1 folds to: every parked lane at once)
    if (parked_mask != 0 && (__ballot(live) == 0 || static_cast<uint32_t>(__popcll(parked_mask)) >= kMergedPark<ALL_OPAQUE>)) {
      if (parked) {
        if (cur != 4u) {
          float v;
          if (ALL_OPAQUE || h.tri == kNoHit) v = h.tri == kNoHit ? 1.0f : 0.0f;  // nothing in the way / (all-opaque) any accepted triangle
          else if (h.opq != 0u) v = 0.0f;                                       // the closest hit is provably opaque (main.comp:325)
          else {
            // the closest hit may be transparent (rare: cloth, glass): the reference's step-by-step walk goes on from this hit AFTER
            // the loop, where its texture code does not sit among the walk's registers; the hit waits in `cold`
            v = -1.0f;
            const uint32_t at = COLD_HIT + 4u * (cur - 1u);
            cold[at] = h.t;
            cold[at + 1] = h.u;
            cold[at + 2] = h.v;
            cold[at + 3] = ubits(h.tri);
          }
          if (cur == 1u) vis0 = v;
          else vis1 = v;
        }
        cur = 0;
        if (todo) start();
      }
    }
    if (__ballot(live) == 0) break;
  }
  if (!ALL_OPAQUE) {
    // light samples whose first hit may be transparent: the rest of ray_scene_intersect_with_transparency (main.comp:306-332) from
    // that hit on, lanes in lock step, then the caller's distance test (main.comp:447 / :487, which looks at the LAST step's hit)
#pragma nounroll
    for (uint32_t j = 0; j < 2u; ++j) {
      float v = j ? vis1 : vis0;
      if (v < 0.0f) {
        const uint32_t at = COLD_HIT + 4u * j;
        Hit sh{cold[at], cold[at + 1], cold[at + 2], fbits(cold[at + 3])};
#if PT_MERGED_COLD
        const vec3 so = cold_get(cold, COLD_ORIGIN), sdir = cold_get(cold, j ? COLD_L1 : COLD_L0);
#else
        const vec3 so = pend.origin, sdir = j ? pend.l1 : pend.l0;
#endif
        Tally t;
        v = shadow_walk<false, false, TEXTURED, true>(sc, sd, tr, so, sdir, sh, t);
        walks.rays += t.rays;
        ext.visits += t.visits;
        ext.tests += t.tests;
        tally.hits += t.hits;
        tally.taps += t.taps;
        if (v != 0.0f && sh.tri != kNoHit) {
          const float correction = lds_load16(stash.quad + 2u * j * stash.quad_stride).w;
          const float limit = lds_load8(stash.pair + j * stash.pair_stride).y;
          if (sh.t + correction < limit) v = 0.0f;
        }
      }
      if (j) vis1 = v;
      else vis0 = v;
    }
  }
}

// The colour update a DEFER vertex left behind (main.comp:536-552), now that its light samples' visibility is known; then the end of
// the sample if the path ended there (main.comp:580-582, 607-612). Same expressions as k_shadow's.
__device__ __forceinline__ void resolve_pending(const pt_dev_state& st, const Stash& stash, vec3 bg_linear, uint32_t slot, Pending& pend,
                                                float vis0, float vis1, PathRegs& p, Tally& tally) {
  if (!(pend.flags & PEND_SHADED)) return;
  vec3 punctual = splat(0.0f), area_light = splat(0.0f);
#pragma unroll
  for (uint32_t j = 0; j < 2u; ++j) {
    const float vis = j ? vis1 : vis0;
    if ((pend.flags & (PEND_S0 << j)) && vis != 0.0f) {
      const pt_f4 a = lds_load16(stash.quad + 2u * j * stash.quad_stride), b = lds_load16(stash.quad + (2u * j + 1u) * stash.quad_stride);
      const float pdf = lds_load8(stash.pair + j * stash.pair_stride).x;
      const vec3 c = mk(a.x, a.y, a.z) * vis * mk(b.x, b.y, b.z) * b.w / pdf;
      if (pend.flags & (PEND_AREA0 << j)) area_light = c;
      else punctual = punctual + c;
      if (pend.flags & (PEND_TAP0 << j)) ++tally.taps;
    }
  }
  vec3 direct = pend.emission;   // main.comp:536
  direct = direct + punctual;    // :537
  direct = direct + area_light;  // :538
  p.color = p.color + direct * pend.af * pend.m;  // :545 / :552 (af == 1 for the opaque branch)
  if (pend.flags & PEND_ENDED) {
    float4 acc = st.accum[slot];
    finish_sample(p.color, p.indirect, bg_linear, acc, tally);
    st.accum[slot] = acc;
    p.indirect = splat(1.0f);
    p.color = splat(0.0f);
  }
  pend.flags = 0;
}

// The path kernel: every queued slot advances by up to `max_vertices` path vertices — extend,
// shade (with its shadow walks), extend, ... — in ONE launch, its ray and path state held in
// registers in between; then the survivors are compacted into the next queue. A pixel's samples
// are sequential (one RNG stream per pixel, Q2), so a slot is the unit of work anyway, and the
// lanes of a wave stay in phase (all traverse, then all shade), which is all the coherence the
// two-kernel wavefront iteration provided once the queue was compacted: measured equal in
// throughput with the state round trip (ray, hit, path: ~220 B per vertex), one of the two
// launches and the per-iteration barrier gone. Same functions, same order per
// slot as k_extend + k_shade, so results are unchanged.
// Queue counters rotate over three words: this launch reads `cnt_in`, appends to `cnt_out` and
// clears `cnt_zero` (the previous launch's input, the next launch's output) — no launch
// clears a word that blocks of the same launch still have to read.
namespace PT_KNS {
template <bool IN_LDS, bool ALL_OPAQUE, int MIN_WAVES, bool PUNCTUAL, bool TEXTURED>
// (launched with 64-, 128- or 256-thread blocks: the bound states the largest; MIN_WAVES = waves per SIMD the register
// allocator must leave room for, which is what sets the occupancy at any of those block sizes)
__global__ __launch_bounds__(256, MIN_WAVES) void k_paths(pt_dev_scene sc, pt_dev_frame f, pt_dev_state st, uint32_t parity,
                                                                   uint32_t cnt_in, uint32_t cnt_out, uint32_t cnt_zero,
                                                                   uint32_t max_vertices, uint32_t xcd_chunk) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = st.counters[cnt_in];
  if (blockIdx.x == 0 && threadIdx.x == 0) st.counters[cnt_zero] = 0;
  Tally ext, tally, walks;
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (f.pc.emissive_triangle_count ? 1u : 0u);
  const vec3 bg_linear = mk(f.bg_linear[0], f.bg_linear[1], f.bg_linear[2]);
  // Workgroups go to the 8 XCDs round-robin (each with its own 4 MiB L2). With `xcd_chunk` = C > 0 (the grid is
  // then a multiple of 8 C) the queue is dealt to the XCDs in chunks of C consecutive workgroups, so an XCD's L2
  // serves neighbouring tiles instead of every eighth one; C = grid / 8 gives each XCD one contiguous eighth.
  uint32_t vblock = blockIdx.x;
  if (xcd_chunk) {
    const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    vblock = ((i / xcd_chunk) * 8u + x) * xcd_chunk + i % xcd_chunk;
  }
  if (vblock * blockDim.x < n) {  // block-uniform
    const Tracer tr = make_tracer<IN_LDS>(sc, lds_dyn);
    const ShadeData sd = make_shade_data<IN_LDS>(sc, tr, lds_dyn);
    for (uint32_t base = vblock * blockDim.x; base < n; base += gridDim.x * blockDim.x) {  // block-uniform trip count
      const uint32_t gid = base + threadIdx.x;
      bool alive = false;
      uint32_t slot = 0;
      if (gid < n) {
        slot = st.queue[parity][gid];
        const float4 o4 = st.ray_o[slot], d4 = st.ray_d[slot];
        const float4 pa = st.path_a[slot], pb = st.path_b[slot];
        const uint32_t ctr = fbits(pb.w);
        PathRegs p{xyz(pa), xyz(pb), fbits(pa.w), ctr_light(ctr), ctr_transp(ctr), ctr_sample(ctr)};
        bool primary = ctr_primary(ctr);
        vec3 o = xyz(o4), d = xyz(d4);
        for (uint32_t k = 0; k < max_vertices; ++k) {  // bounded: every wave reaches the end of the grid
          Hit h{kInfinity, 0.0f, 0.0f, kNoHit};
          traverse<IN_LDS, false>(tr, o, d, h, ext);
          if (primary) st.primary_id[slot] = h.tri;
          primary = false;
          vec3 next_o = mk(0, 0, 0), next_d = mk(0, 0, 0);
          bool shaded;
#ifdef PT_DIAG_LANES  // diagnostic build: texture_taps counts the wave-level calls of shade_vertex (lanes per call = rays_closest / taps)
          if (mbcnt(__ballot(true)) == 0) ++tally.taps;
#endif
          alive = shade_vertex<IN_LDS, ALL_OPAQUE, false, PUNCTUAL, TEXTURED>(sc, sd, f, st, tr, n_lights, bg_linear, slot, h, d, p,
                                                                              next_o, next_d, tally, walks, shaded);
          if (!alive) break;
          o = next_o;
          d = next_d;
        }
        if (alive) {
          st.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.0f);
          st.ray_d[slot] = make_float4(d.x, d.y, d.z, 0.0f);
        }
        st.path_a[slot] = make_float4(p.indirect.x, p.indirect.y, p.indirect.z, ubits(p.rng));
        st.path_b[slot] = make_float4(p.color.x, p.color.y, p.color.z, ubits(ctr_pack(p.light_steps, p.transp_steps, primary, p.sample)));
      }
      const AppendSlots at = block_append2(alive, &st.counters[cnt_out], false, nullptr);
      if (alive) st.queue[parity ^ 1u][at.a] = slot;
    }
  }
  flush_tally(ext, st, PT_K_EXTEND);
  flush_tally(tally, st, PT_K_SHADE);
  flush_tally(walks, st, PT_K_SHADOW);
}

#ifndef PT_FAST_ARITH  // (the fast-arithmetic compilation is for LDS-resident scenes: pt_context.hip)
// k_paths for scenes read from global memory with at most two light samples per vertex: same state, same queues, same per-slot
// order of work, but a vertex's shadow walks ride with the next vertex's closest-hit walk in one loop (merged_walk). The last
// vertex a slot shades in a launch leaves light samples behind: one more round walks those alone, so nothing pends across launches.
// SAMPLES: light samples per vertex the stash has room for (40 bytes each). One-sample scenes (a sun and nothing else: cfg4) get a
// kernel of their own for all-opaque untextured scenes, whose LDS then holds 16 stack rows where two samples leave room for 9;
// a compile-time parameter because the two-sample kernels' register allocation does not survive ANY change to this address
// arithmetic (a sample-major layout without the parameter: cfg3 2048 -> 1239 Mrays/s, the pair record's tuple spill back).
template <bool ALL_OPAQUE, int MIN_WAVES, bool TEXTURED, bool WIDE = false, uint32_t SAMPLES = 2>
__global__ __launch_bounds__(256, MIN_WAVES) void k_paths_merged(pt_dev_scene sc, pt_dev_frame f, pt_dev_state st, uint32_t parity,
                                                                          uint32_t cnt_in, uint32_t cnt_out, uint32_t cnt_zero,
                                                                          uint32_t max_vertices, uint32_t xcd_chunk) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = st.counters[cnt_in];
  if (blockIdx.x == 0 && threadIdx.x == 0) st.counters[cnt_zero] = 0;
  Tally ext, tally, walks;
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (f.pc.emissive_triangle_count ? 1u : 0u);
  const vec3 bg_linear = mk(f.bg_linear[0], f.bg_linear[1], f.bg_linear[2]);
  uint32_t vblock = blockIdx.x;
  if (xcd_chunk) {
    const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    vblock = ((i / xcd_chunk) * 8u + x) * xcd_chunk + i % xcd_chunk;
  }
  if (vblock * blockDim.x < n) {  // block-uniform
    const Tracer tr = make_tracer<false>(sc, lds_dyn);
    const ShadeData sd = make_shade_data<false>(sc, tr, lds_dyn);
    // LDS: [stack rows x blockDim words | stash: 2 quads per light sample x blockDim | a pair per light sample x blockDim]
    const uint32_t rows = sc.lds_stack_levels ? sc.lds_stack_levels : sc.stack_entries;
    const uint32_t stash_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_dyn + static_cast<size_t>(blockDim.x) * rows));
    const Stash stash{stash_base + 16u * threadIdx.x, 16u * blockDim.x, stash_base + 32u * SAMPLES * blockDim.x + 8u * threadIdx.x, 8u * blockDim.x};
    for (uint32_t base = vblock * blockDim.x; base < n; base += gridDim.x * blockDim.x) {  // block-uniform trip count
      const uint32_t gid = base + threadIdx.x;
      bool alive = false;
      uint32_t slot = 0;
      if (gid < n) {
        slot = st.queue[parity][gid];
        const float4 o4 = st.ray_o[slot], d4 = st.ray_d[slot];
        const float4 pa = st.path_a[slot], pb = st.path_b[slot];
        const uint32_t ctr = fbits(pb.w);
        PathRegs p{xyz(pa), xyz(pb), fbits(pa.w), ctr_light(ctr), ctr_transp(ctr), ctr_sample(ctr)};
        bool primary = ctr_primary(ctr);
        vec3 o = xyz(o4), d = xyz(d4);
        Pending pend;
        pend.origin = pend.l0 = pend.l1 = pend.emission = pend.m = splat(0.0f);
        pend.af = 1.0f;
        float vis0 = 0.0f, vis1 = 0.0f;
        bool has_ray = true;
        volatile float cold[COLD_WORDS];
        for (uint32_t k = 0;; ++k) {  // bounded: at most max_vertices + 1 rounds, every wave reaches the end of the grid
#if PT_MERGED_COLD
          // park what the walk does not need (see merged_walk) ...
          cold_put(cold, COLD_INDIRECT, p.indirect);
          cold_put(cold, COLD_COLOR, p.color);
          cold[COLD_RNG] = ubits(p.rng);
          cold[COLD_CTR] = ubits(ctr_pack(p.light_steps, p.transp_steps, false, p.sample));
          cold_put(cold, COLD_EMISSION, pend.emission);
          cold_put(cold, COLD_M, pend.m);
          cold[COLD_AF] = pend.af;
          cold_put(cold, COLD_ORIGIN, pend.origin);
          cold_put(cold, COLD_L0, pend.l0);
          cold_put(cold, COLD_L1, pend.l1);
          cold_put(cold, COLD_RAY_O, o);
          cold_put(cold, COLD_RAY_D, d);
#endif
          Hit h{kInfinity, 0.0f, 0.0f, kNoHit};
          merged_walk<ALL_OPAQUE, TEXTURED, WIDE>(sc, sd, tr, stash, pend, has_ray, o, d, cold, h, vis0, vis1, ext, walks, tally);
#if PT_MERGED_COLD
          // ... and take it back
          p.indirect = cold_get(cold, COLD_INDIRECT);
          p.color = cold_get(cold, COLD_COLOR);
          p.rng = fbits(cold[COLD_RNG]);
          {
            const uint32_t c = fbits(cold[COLD_CTR]);
            p.light_steps = ctr_light(c);
            p.transp_steps = ctr_transp(c);
            p.sample = ctr_sample(c);
          }
          pend.emission = cold_get(cold, COLD_EMISSION);
          pend.m = cold_get(cold, COLD_M);
          pend.af = cold[COLD_AF];
          d = cold_get(cold, COLD_RAY_D);
#endif
          resolve_pending(st, stash, bg_linear, slot, pend, vis0, vis1, p, tally);
          if (!has_ray) break;
          if (primary) st.primary_id[slot] = h.tri;
          primary = false;
          vec3 next_o = mk(0, 0, 0), next_d = mk(0, 0, 0);
          bool shaded;
          alive = shade_vertex<false, ALL_OPAQUE, false, true, TEXTURED, true>(sc, sd, f, st, tr, n_lights, bg_linear, slot, h, d, p, next_o,
                                                                              next_d, tally, walks, shaded, &pend, &stash);
          if (alive) {
            o = next_o;
            d = next_d;
          }
          if (!alive || k + 1 >= max_vertices) {  // no further ray in this launch: one more round if light samples are left over
            has_ray = false;
            if (!pend.flags) break;
          }
        }
        if (alive) {
          st.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.0f);
          st.ray_d[slot] = make_float4(d.x, d.y, d.z, 0.0f);
        }
        st.path_a[slot] = make_float4(p.indirect.x, p.indirect.y, p.indirect.z, ubits(p.rng));
        st.path_b[slot] = make_float4(p.color.x, p.color.y, p.color.z, ubits(ctr_pack(p.light_steps, p.transp_steps, primary, p.sample)));
      }
      const AppendSlots at = block_append2(alive, &st.counters[cnt_out], false, nullptr);
      if (alive) st.queue[parity ^ 1u][at.a] = slot;
    }
  }
  flush_tally(ext, st, PT_K_EXTEND);
  flush_tally(tally, st, PT_K_SHADE);
  flush_tally(walks, st, PT_K_SHADOW);
}

#endif  // !PT_FAST_ARITH

}  // namespace PT_KNS
using namespace PT_KNS;

#ifndef PT_FAST_ARITH
namespace PT_KNS {  // kernels live in the build's own namespace: the same template compiled twice must not share a symbol
// Column sums of the per-block counter table: one workgroup per kernel slice, so the host reads
// PT_K_COUNT x PT_ST_WORDS words instead of the whole table (16 MB at 1080p: ~1.5 ms of the frame's wall time).
__global__ __launch_bounds__(256) void k_sum_stats(pt_dev_state st) {
  __shared__ unsigned long long part[4][PT_ST_WORDS];
  const unsigned long long* rows = st.stats + static_cast<size_t>(blockIdx.x) * st.stat_rows * PT_ST_WORDS;
  unsigned long long acc[PT_ST_WORDS] = {};
  for (uint32_t r = threadIdx.x; r < st.stat_rows; r += blockDim.x)
    for (int w = 0; w < PT_ST_WORDS; ++w) acc[w] += rows[static_cast<size_t>(r) * PT_ST_WORDS + w];
  for (int w = 0; w < PT_ST_WORDS; ++w) {
    unsigned long long v = acc[w];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane_id() == 0) part[wave_id()][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < PT_ST_WORDS)
    st.stat_totals[blockIdx.x * PT_ST_WORDS + threadIdx.x] =
        part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// slot order -> pixel order: accum_out = (sum R, sum G, sum B, valid samples)
__global__ __launch_bounds__(256) void k_finalize(pt_dev_frame f, pt_dev_state st, float4* accum_out, uint32_t* primary_out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x, y;
  if (slot >= f.slot_count || !slot_to_pixel(f, slot, x, y)) return;
  const uint32_t index = y * f.pc.resolution_x + x;
  const float4 a = st.accum[slot];
  const float valid = static_cast<float>(f.sample_end - f.sample_begin) - a.w;
  accum_out[index] = make_float4(a.x, a.y, a.z, valid);
  if (primary_out) primary_out[index] = st.primary_id[slot];
}

__device__ __forceinline__ uint32_t to_unorm8(float x) {  // imageStore to rgba8: NaN -> 0, round to nearest even
  if (!(x > 0.0f)) return 0u;
  if (x >= 1.0f) return 255u;
  return static_cast<uint32_t>(__builtin_rintf(x * 255.0f));
}

// main():616-619
__global__ __launch_bounds__(256) void k_resolve(const float4* __restrict__ accum, uint32_t pixel_count, uint32_t* __restrict__ rgba8) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixel_count) return;
  const float4 a = accum[i];
  const uint32_t r = to_unorm8(pt_pow(a.x / a.w, kInvGamma));
  const uint32_t g = to_unorm8(pt_pow(a.y / a.w, kInvGamma));
  const uint32_t b = to_unorm8(pt_pow(a.z / a.w, kInvGamma));
  rgba8[i] = r | (g << 8) | (b << 16) | (255u << 24);
}

}  // namespace PT_KNS
#endif  // !PT_FAST_ARITH

// =========================================================================== launchers

namespace PT_LNS {

static inline uint32_t blocks_for(uint32_t n, uint32_t block) { return n ? (n + block - 1) / block : 1u; }

#ifndef PT_FAST_ARITH
void launch_init(const pt_dev_frame& f, const pt_dev_state& st, hipStream_t s) {
  hipLaunchKernelGGL(k_init, dim3(blocks_for(f.slot_count, 256)), dim3(256), 0, s, f, st);
}

#endif

static inline size_t trace_lds_bytes(const pt_dev_scene& sc, uint32_t block) {
  if (!sc.scene_in_lds) return static_cast<size_t>(block) * (sc.lds_stack_levels ? sc.lds_stack_levels : sc.stack_entries) * sizeof(uint32_t);
  return static_cast<size_t>(block) * sc.stack_entries * sizeof(uint32_t) + 64u * static_cast<size_t>(sc.pair_count) +
         48u * static_cast<size_t>(sc.triangle_count);
}

static inline size_t shade_lds_bytes(const pt_dev_scene& sc, uint32_t block) {
  return trace_lds_bytes(sc, block) + (sc.scene_in_lds ? 16u * static_cast<size_t>(shade_lds_float4s(sc)) : 0u);
}

#ifndef PT_FAST_ARITH
void launch_extend(const pt_dev_scene& sc, const pt_dev_state& st, uint32_t parity, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s) {
  uint32_t grid = blocks_for(max_entries, block);
  if (grid > max_blocks) grid = max_blocks;
  if (sc.scene_in_lds) hipLaunchKernelGGL(k_extend<true>, dim3(grid), dim3(block), trace_lds_bytes(sc, block), s, sc, st, parity);
  else hipLaunchKernelGGL(k_extend<false>, dim3(grid), dim3(block), trace_lds_bytes(sc, block), s, sc, st, parity);
}

void launch_shadow(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s) {
  uint32_t grid = blocks_for(max_entries, block);
  if (grid > max_blocks) grid = max_blocks;
  if (sc.all_opaque) hipLaunchKernelGGL(k_shadow<true>, dim3(grid), dim3(block), trace_lds_bytes(sc, block), s, sc, f, st);
  else hipLaunchKernelGGL(k_shadow<false>, dim3(grid), dim3(block), trace_lds_bytes(sc, block), s, sc, f, st);
}

void launch_shade(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity,
                  uint32_t max_entries, uint32_t block, uint32_t max_blocks, bool has_lights, bool split, bool compact,
                  hipStream_t s) {
  uint32_t grid = blocks_for(max_entries, block);
  if (grid > max_blocks) grid = max_blocks;
  const size_t lds = (has_lights || sc.scene_in_lds) ? shade_lds_bytes(sc, block) : 0;
  const int variant = (sc.scene_in_lds ? 2 : 0) | (sc.all_opaque ? 1 : 0);
  if (split) {  // global scene with lights: requests out, walks in k_shadow
    if (sc.all_opaque) hipLaunchKernelGGL((k_shade<false, true, true>), dim3(grid), dim3(block), 0, s, sc, f, st, parity, compact ? 1u : 0u);
    else hipLaunchKernelGGL((k_shade<false, false, true>), dim3(grid), dim3(block), 0, s, sc, f, st, parity, compact ? 1u : 0u);
    return;
  }
  switch (variant) {
    case 3: hipLaunchKernelGGL((k_shade<true, true, false>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, compact ? 1u : 0u); break;
    case 2: hipLaunchKernelGGL((k_shade<true, false, false>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, compact ? 1u : 0u); break;
    case 1: hipLaunchKernelGGL((k_shade<false, true, false>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, compact ? 1u : 0u); break;
    default: hipLaunchKernelGGL((k_shade<false, false, false>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, compact ? 1u : 0u); break;
  }
}

#endif

void launch_paths(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                  uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                  uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s) {
  uint32_t grid = blocks_for(max_entries, block);
  if (grid > max_blocks) grid = max_blocks;
  if (xcd_chunk) {  // whole rounds of 8 chunks
    if (xcd_chunk > (grid + 7u) / 8u) xcd_chunk = (grid + 7u) / 8u;
    const uint32_t round = 8u * xcd_chunk;
    grid = (grid + round - 1u) / round * round;
  }
  const uint32_t remap = xcd_chunk;
  const size_t lds = shade_lds_bytes(sc, block);
  const int variant = (sc.scene_in_lds ? 2 : 0) | (sc.all_opaque ? 1 : 0);
  // occupancy bound by scene class (swept): 4 waves/SIMD while the scene sits in LDS or in an XCD's L2
  // (VALU bound: spills cost more than a 5th wave hides), 5 once its fetches go to the Infinity Cache / HBM
  const bool big = !sc.scene_in_lds && 64ull * sc.pair_count + 48ull * sc.triangle_count > (4ull << 20);
  // Kernels specialised on what the scene provably lacks (dead code and its registers go: cfg2 +4.8 % without the
  // punctual-light arms; LDS scenes only — the others wait on memory): no directional or point light -> the light
  // loop is the area sample alone; no material names a texture -> no sampling code.
  const bool punctual = sc.directional_light_count + sc.point_light_count != 0;
  const bool textured = sc.any_texture != 0;
#define PT_LAUNCH_PATHS(A, B, W, P, T) \
  hipLaunchKernelGGL((k_paths<A, B, W, P, T>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, remap)
#define PT_LAUNCH_PATHS_LDS(B)                                                     \
  do {                                                                             \
    if (textured) PT_LAUNCH_PATHS(true, B, PT_PATHS_MIN_WAVES, true, true);        \
    else if (punctual) PT_LAUNCH_PATHS(true, B, PT_PATHS_MIN_WAVES, true, false);  \
    else PT_LAUNCH_PATHS(true, B, PT_PATHS_MIN_WAVES_PLAIN, false, false);         \
  } while (0)
  // (scenes in global memory: no texture -> the kernel without the six inlined copies of the sampler: half the code (57 -> ~30 KB;
  // the instruction cache is 64 KB per two CUs and the generic kernel's waves wait for instructions 10 % of their time) and
  // fewer registers to spill)
#define PT_LAUNCH_PATHS_GLOBAL(B)                                                            \
  do {                                                                                       \
    if (big && textured) PT_LAUNCH_PATHS(false, B, PT_PATHS_MIN_WAVES_GLOBAL, true, true);   \
    else if (big) PT_LAUNCH_PATHS(false, B, PT_PATHS_MIN_WAVES_GLOBAL, true, false);         \
    else if (textured) PT_LAUNCH_PATHS(false, B, PT_PATHS_MIN_WAVES, true, true);            \
    else PT_LAUNCH_PATHS(false, B, PT_PATHS_MIN_WAVES, true, false);                         \
  } while (0)
  switch (variant) {
    case 3: PT_LAUNCH_PATHS_LDS(true); break;
    case 2: PT_LAUNCH_PATHS_LDS(false); break;
    case 1: PT_LAUNCH_PATHS_GLOBAL(true); break;
    default: PT_LAUNCH_PATHS_GLOBAL(false); break;
  }
#undef PT_LAUNCH_PATHS_GLOBAL
#undef PT_LAUNCH_PATHS_LDS
#undef PT_LAUNCH_PATHS
}

#ifndef PT_FAST_ARITH
// k_paths_merged: scenes read from global memory, at most two light samples per vertex (the caller checks both). LDS per thread:
// the traversal-stack rows + 40 bytes of stash per light sample of a vertex the kernel has room for.
uint32_t merged_stash_samples(const pt_dev_scene& sc, uint32_t n_lights) { return (n_lights <= 1u && sc.all_opaque && !sc.any_texture) ? 1u : 2u; }
size_t merged_lds_bytes(const pt_dev_scene& sc, uint32_t block, uint32_t n_lights) {
  return trace_lds_bytes(sc, block) + 40u * merged_stash_samples(sc, n_lights) * static_cast<size_t>(block);
}

void launch_paths_merged(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                         uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                         uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s) {
  uint32_t grid = blocks_for(max_entries, block);
  if (grid > max_blocks) grid = max_blocks;
  if (xcd_chunk) {  // whole rounds of 8 chunks
    if (xcd_chunk > (grid + 7u) / 8u) xcd_chunk = (grid + 7u) / 8u;
    const uint32_t round = 8u * xcd_chunk;
    grid = (grid + round - 1u) / round * round;
  }
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (f.pc.emissive_triangle_count ? 1u : 0u);
  const size_t lds = merged_lds_bytes(sc, block, n_lights);
  const bool textured = sc.any_texture != 0;
  const bool one_sample = merged_stash_samples(sc, n_lights) == 1u;
#if PT_STACK_CHILD && !defined(PT_FAST_ARITH)  // the wide4 walk exists in the exact compilation only (its entries are packed children)
#define PT_LAUNCH_MERGED(B, W, T)                                                                                                                \
  do {                                                                                                                                           \
    if (sc.wide) hipLaunchKernelGGL((k_paths_merged<B, W, T, true>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, xcd_chunk); \
    else hipLaunchKernelGGL((k_paths_merged<B, W, T, false>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, xcd_chunk);        \
  } while (0)
#else
#define PT_LAUNCH_MERGED(B, W, T) \
  hipLaunchKernelGGL((k_paths_merged<B, W, T>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, xcd_chunk)
#endif
#define PT_LAUNCH_MERGED_BY_SIZE(B)                                      \
  do {                                                                   \
    if (textured) PT_LAUNCH_MERGED(B, PT_MERGED_MIN_WAVES, true);        \
    else PT_LAUNCH_MERGED(B, PT_MERGED_MIN_WAVES, false);                \
  } while (0)
  if (one_sample) {  // all-opaque, untextured, one light sample per vertex (merged_stash_samples)
#if PT_STACK_CHILD && !defined(PT_FAST_ARITH)
    if (sc.wide) hipLaunchKernelGGL((k_paths_merged<true, PT_MERGED_MIN_WAVES, false, true, 1u>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, xcd_chunk);
    else
#endif
      hipLaunchKernelGGL((k_paths_merged<true, PT_MERGED_MIN_WAVES, false, false, 1u>), dim3(grid), dim3(block), lds, s, sc, f, st, parity, cnt_in, cnt_out, cnt_zero, max_vertices, xcd_chunk);
  } else if (sc.all_opaque) PT_LAUNCH_MERGED_BY_SIZE(true);
  else PT_LAUNCH_MERGED_BY_SIZE(false);
#undef PT_LAUNCH_MERGED_BY_SIZE
#undef PT_LAUNCH_MERGED
}

#endif  // !PT_FAST_ARITH

#ifndef PT_FAST_ARITH
void launch_sum_stats(const pt_dev_state& st, hipStream_t s) { hipLaunchKernelGGL(k_sum_stats, dim3(PT_K_COUNT), dim3(256), 0, s, st); }

void launch_finalize(const pt_dev_frame& f, const pt_dev_state& st, float4* accum_out, uint32_t* primary_out, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize, dim3(blocks_for(f.slot_count, 256)), dim3(256), 0, s, f, st, accum_out, primary_out);
}

void launch_resolve(const float4* accum, uint32_t pixel_count, uint32_t* rgba8, hipStream_t s) {
  hipLaunchKernelGGL(k_resolve, dim3(blocks_for(pixel_count, 256)), dim3(256), 0, s, accum, pixel_count, rgba8);
}

#endif

}  // namespace PT_LNS
