// Device-side data layout shared by the kernels (pt_kernels.hip) and the context that
// owns the HBM allocations (pt_context.hip).  See DESIGN.md "Data layout in HBM".
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pathtracer_amd.h"

// Rows of a lane's traversal stack beyond its LDS rows (private memory; traverse() in pt_kernels.hip). The host only chooses a short
// LDS stack when the tree's depth fits LDS rows + these.
#define PT_DEEP_LEVELS 48u

// counters[] words
enum { PT_CNT_QUEUE0 = 0, PT_CNT_QUEUE1 = 1, PT_CNT_QUEUE2 = 2, PT_CNT_SHADOW = 3, PT_CNT_WORDS = 4 };
// stats[] words (unsigned long long), order of flush_tally()
enum { PT_ST_PATHS = 0, PT_ST_RAYS, PT_ST_VISITS, PT_ST_TESTS, PT_ST_HITS, PT_ST_TAPS, PT_ST_VERTS, PT_ST_NANS, PT_ST_WORDS };
// one block of PT_ST_WORDS per kernel
enum { PT_K_INIT = 0, PT_K_EXTEND, PT_K_SHADE, PT_K_SHADOW, PT_K_COUNT, PT_K_PATHS = PT_K_COUNT /* timing slot only */ };

// The scene in HBM, split by who reads it:
//   pairs      traversal only   — one 64-byte record per INNER node holding both children
//                                  {min, ref | max, count} x 2, breadth-first order (derived
//                                  at upload from the reference's PackedBVHNode array; the
//                                  boxes and triangle ranges are copied bit for bit); small
//                                  scenes are staged whole in LDS by every traversal workgroup
//   tri_pos    traversal + shade — the reference's binding 0 verbatim: 3 x vec4 per triangle
//   tri_attr   shade only        — 64 bytes per triangle: 3 normals, 3 uvs, material index
//                                  (the reference's 192-byte vertex triple minus the
//                                  tangent / bitangent / duplicate position it never reads, Q15)
//   materials  shade             — the reference's 64-byte Material array verbatim
struct pt_dev_scene {
  const float4* pairs;
  const float4* wide;                // PT_BVH=wide4 (opt-in): 64-byte records of up to four children with 8-bit boxes (pt_kernels.hip); NULL otherwise
  const float4* tri_pos;
  const float4* tri_attr;
  const float4* materials;
  const float4* directional_lights;  // 2 x float4 per light
  const float4* point_lights;
  const uint32_t* emissive_triangles;
  const uint4* textures;             // {texel offset, width, height, mag_filter | wrap_s << 2 | wrap_t << 4}
  const uint32_t* texels;            // all images, RGBA8 packed little-endian
  uint32_t directional_light_count;
  uint32_t point_light_count;
  uint32_t stack_entries;            // LDS stack rows per lane (bvh_max_depth + 1)
  uint32_t triangle_count;
  uint32_t material_count;
  uint32_t emissive_count;
  uint32_t pair_count;
  uint32_t wide_count;               // wide records
  uint32_t wide_stack_need;          // deferred children a wide walk may hold at once (worst case over the tree)
  uint32_t root_ref, root_cnt;       // the root as a child reference: pair 0 (count 0) or a leaf's triangle range
  uint32_t scene_in_lds;             // 1: the scene (pair records, triangle positions; for shading also attributes, materials, lights, emitter list) is staged whole in the workgroup's LDS
  uint32_t all_opaque;               // 1: no surface can report alpha != 1 (shadow rays may stop at the first hit)
  uint32_t any_texture;              // 0: no material names a texture (kernels without sampling code)
  uint32_t packed_stack;             // scenes in global memory: a traversal-stack entry can hold a child as (ref << 7 | cnt): the ptd kernels; else ptd_rs
  uint32_t lds_stack_levels;         // scenes in global memory: traversal-stack rows kept in LDS per lane; deeper entries overflow to
                                     // private memory (0: all stack_entries rows in LDS). Set per render by the host.
};

struct pt_dev_frame {
  pt_push_constants pc;
  float bg_linear[3];      // pow(bg_color_srgb, GAMMA), main.comp:506
  uint32_t sqrt_samples;   // int(sqrt(sample_count)), main.comp:596
  uint32_t sample_begin;   // strata [begin, end) rendered by this rank
  uint32_t sample_end;
  uint32_t rank, num_ranks;
  uint32_t slot_base;      // first slot of the band being rendered (slots are 8x8-pixel tiles, row-major)
  uint32_t slot_count;     // slots in this band (a multiple of 64)
  uint32_t tiles_x;
};

// Per-slot path state, SoA of 16-byte words (one coalesced dwordx4 access per field per wave)
struct pt_dev_state {
  float4* ray_o;       // origin.xyz
  float4* ray_d;       // direction.xyz
  uint4* hit;          // triangle, u, v, t
  float4* path_a;      // throughput (`indirect`).xyz, rng state
  float4* path_b;      // radiance (`color`).xyz, packed step counters + stratum
  float4* accum;       // sum of valid samples .xyz, dropped-sample count
  uint32_t* primary_id;
  uint32_t* queue[2];
  // split pipeline (scenes in global memory with lights): per-slot deferred direct-light update
  float4* pend;           // [5][slot_stride]: emission|alpha factor, throughput|flags, shadow origin, colour, end throughput
  float4* req;            // [n_lights][4][slot_stride]: l|limit, factor|correction, brdf|n.l, pdf|emitter|tap
  uint32_t* queue_shadow; // slots whose vertex k_shadow has to finish
  uint32_t slot_stride;
  uint32_t* counters;
  unsigned long long* stats;  // [kernel][stat_rows][PT_ST_WORDS]: one row per block index, no atomics
  unsigned long long* stat_totals;  // [kernel][PT_ST_WORDS]: column sums of `stats`, formed by k_sum_stats after the frame
  uint32_t stat_rows;         // >= the largest grid (slot_count / 64 + 1)
};

namespace ptd {
void launch_init(const pt_dev_frame& f, const pt_dev_state& st, hipStream_t s);
void launch_extend(const pt_dev_scene& sc, const pt_dev_state& st, uint32_t parity, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s);
void launch_shade(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity,
                  uint32_t max_entries, uint32_t block, uint32_t max_blocks, bool has_lights, bool split, bool compact,
                  hipStream_t s);
void launch_shadow(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s);
void launch_paths(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                  uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                  uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s);
void launch_paths_merged(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                         uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                         uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s);
uint32_t merged_stash_samples(const pt_dev_scene& sc, uint32_t n_lights);  // light samples per vertex the launched k_paths_merged has stash for (1 or 2)
size_t merged_lds_bytes(const pt_dev_scene& sc, uint32_t block, uint32_t n_lights);
void launch_sum_stats(const pt_dev_state& st, hipStream_t s);
void launch_finalize(const pt_dev_frame& f, const pt_dev_state& st, float4* accum_out, uint32_t* primary_out, hipStream_t s);
void launch_resolve(const float4* accum, uint32_t pixel_count, uint32_t* rgba8, hipStream_t s);
}  // namespace ptd

namespace ptd_rs {  // pt_kernels_rs.hip: the traversal kernels with (record, side) stack entries (scenes that fail pt_dev_scene::packed_stack)
void launch_extend(const pt_dev_scene& sc, const pt_dev_state& st, uint32_t parity, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s);
void launch_shade(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity,
                  uint32_t max_entries, uint32_t block, uint32_t max_blocks, bool has_lights, bool split, bool compact,
                  hipStream_t s);
void launch_shadow(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t max_entries, uint32_t block,
                   uint32_t max_blocks, hipStream_t s);
void launch_paths(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                  uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                  uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s);
void launch_paths_merged(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                         uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                         uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s);
uint32_t merged_stash_samples(const pt_dev_scene& sc, uint32_t n_lights);  // light samples per vertex the launched k_paths_merged has stash for (1 or 2)
size_t merged_lds_bytes(const pt_dev_scene& sc, uint32_t block, uint32_t n_lights);
}  // namespace ptd_rs

namespace ptd_fast {  // pt_kernels_fast.hip: k_paths with the shading half on hardware-approximate arithmetic (PT_ARITH=fast)
void launch_paths(const pt_dev_scene& sc, const pt_dev_frame& f, const pt_dev_state& st, uint32_t parity, uint32_t cnt_in,
                  uint32_t cnt_out, uint32_t cnt_zero, uint32_t max_entries, uint32_t block, uint32_t max_blocks,
                  uint32_t max_vertices, uint32_t xcd_chunk, hipStream_t s);
}  // namespace ptd_fast
