/*
 * pt_oracle.h — CPU ORACLE (test infrastructure only; see pt_oracle.c header).
 * Uses the public C-ABI structs of include/pathtracer_amd.h for its inputs so that a
 * test hands byte-identical data to the oracle and to the HIP path.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/pathtracer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* build_bvh, src/bvh/bvh.cc:161-169 */
int orc_build_bvh(const pt_vertex* vertices, uint32_t vertex_count, uint32_t* triangles,
                  uint32_t triangle_count, uint32_t bin_count, pt_bvh_node* nodes_out,
                  uint32_t* node_count_out, uint32_t* depth_out);

/* RenderJob::process_scene, src/renderer/render_job.cc:76-121 */
int orc_process_scene(const pt_vertex* vertices, uint32_t vertex_count, const uint32_t* triangles,
                      uint32_t triangle_count, const pt_material* materials, uint32_t material_count,
                      const pt_bvh_node* nodes, uint32_t node_count, float* vertex_positions_out,
                      pt_packed_vertex* packed_vertex_out, uint32_t* emissive_out,
                      uint32_t* emissive_count_out, pt_packed_bvh_node* packed_nodes_out);

/* PushConstants ctor, src/renderer/render_job.cc:836-873 */
int orc_make_push_constants(const pt_render_config* config, const pt_camera* camera,
                            uint32_t emissive_triangle_count, uint32_t seed, pt_push_constants* out);

/* main.comp:587-620 for the pixel rows [y_begin, y_end), num_threads host threads.
 * Outputs are full-image buffers indexed y*resolution_x+x; rows outside the range are
 * left untouched. Any output may be NULL. */
int orc_render(const pt_scene_desc* scene, const pt_render_params* params, uint8_t* rgba8_out,
               float* accum_out, uint32_t* primary_id_out, pt_stats* stats_out, int num_threads,
               uint32_t y_begin, uint32_t y_end);

/* Same, with checker options. ORC_FLAG_ANYHIT_OPAQUE_SHADOW mirrors the HIP path's shadow-ray
 * shortcut (first accepted hit ends the traversal when the scene is provably all-opaque):
 * images are identical to orc_render by construction, only node_visits / tri_tests /
 * hits_shaded differ; it exists so the GPU work counters can be checked for equality. */
#define ORC_FLAG_ANYHIT_OPAQUE_SHADOW 1u
/* ORC_FLAG_CULL_BEHIND mirrors the HIP path's extra box cull (t_far < 0: the box lies behind
 * the ray origin): no triangle in such a box can pass `t > epsilon`, so images
 * are identical (tests/test_oracle_shader.py checks it); visits/tests shrink. */
#define ORC_FLAG_CULL_BEHIND 2u
/* ORC_FLAG_SELF_FIRST mirrors the HIP path's second shadow-ray shortcut (with ORC_FLAG_ANYHIT_OPAQUE_SHADOW on an all-opaque scene):
 * a shadow query first tests the triangle the shading point lies on. A light behind the surface sends the ray, which starts
 * NORMAL_DELTA above that triangle, straight through it: one triangle test decides what the walk would find after a full descent
 * (scripts/self_first_model.py: 51 % of the 5 M-triangle rocks' shadow rays). Any accepted triangle decides, so images are identical. */
#define ORC_FLAG_SELF_FIRST 4u
/* ORC_FLAG_ANYHIT_SPAN_ORDER mirrors the order in which the HIP path's any-hit walks enter two children that are both hit: the one
 * in whose box the ray stays longer (t_far - max(t_near, 0)) first, the left one on a tie. A yes / no query's answer does not depend
 * on the order (only closest-hit walks owe the reference theirs, Q14); the work does (scripts/anyhit_order_model.py: -8.6 % of all
 * node visits on the rocks, -10.7 % on the 1 M-triangle soup). */
#define ORC_FLAG_ANYHIT_SPAN_ORDER 8u
/* ORC_FLAG_OPAQUE_TRIANGLES mirrors the HIP path's shortcut for scenes that are NOT all-opaque: when the closest hit of a shadow-walk
 * step lies on a triangle whose material is provably opaque (base_color_factor.a == 1 and no base-colour texture or one whose alpha
 * bytes are all 255), the walk returns 0 (main.comp:325) without get_interpolated_data, the material fetch and the texture tap: the
 * same result with fewer hits_shaded / texture_taps. */
#define ORC_FLAG_OPAQUE_TRIANGLES 16u
/* ORC_FLAG_STALE_SKIP: a closest-hit walk of a path ray does not visit a deferred INNER node whose entry distance, noted when the
 * node was deferred, is no longer below the closest hit. The image cannot change (the node's children lie inside its box and the
 * slab arithmetic is monotonic in the bounds, so both would be culled); node_visits shrinks by 4-5 %. A MODEL and the checker
 * mirror of a kernel that was built in round 3 and measured slower: no shipped HIP kernel sets it. */
#define ORC_FLAG_STALE_SKIP 32u
/* ORC_FLAG_TRANSPARENT_PROBE (scenes that are NOT all-opaque): the triangles that are not provably opaque get a small tree of their
 * own; a shadow ray that none of them accepts (any-hit walk of that tree, counted in node_visits / tri_tests, not in rays) can only
 * meet opaque surfaces, so the query is the all-opaque one — the emitter triangle first, with ORC_FLAG_SELF_FIRST the shading point's
 * own triangle, then an any-hit walk. A ray that one of them does accept takes the reference's walk. Same image by construction;
 * node visits x 0.53 on cfg3. A MODEL and the checker mirror of kernels that were built in round 3 (three forms, all bit-identical to
 * this mode, counters included) and measured 6-16 % slower on cfg3: no shipped HIP kernel sets it. */
#define ORC_FLAG_TRANSPARENT_PROBE 64u
int orc_render_ex(const pt_scene_desc* scene, const pt_render_params* params, uint8_t* rgba8_out,
                  float* accum_out, uint32_t* primary_id_out, pt_stats* stats_out, int num_threads,
                  uint32_t y_begin, uint32_t y_end, uint32_t flags);
int orc_scene_all_opaque(const pt_scene_desc* scene);

/* main.comp:616-619 for one pixel's accumulator */
void orc_resolve_pixel(float r, float g, float b, float valid, uint8_t* out_rgba);

/* Per-function exports for the independent pins (tests/test_oracle_functions.py): main.comp:169-185, 189-217,
 * 336-366, 146-156, 370-405. Vectors are 3 floats; *state is the RNG state before and after. */
int orc_ray_aabb(const float* origin, const float* dir, const float* bmin, const float* bmax, float* t_near);
int orc_ray_triangle(const float* origin, const float* dir, const float* tri9, float t_in, float* t, float* u, float* v);
float orc_distribution_ggx(float n_dot_h, float a2);
float orc_geometry_smith1(const float* n, const float* h, const float* x, float a2);
void orc_microfacet_brdf(const float* n, const float* v, const float* l, float metal, float a2, const float* base_color, float* out);
void orc_orthogonal_matrix(const float* w, float* out9);
float orc_rand_float(uint32_t* state);
void orc_uniform_hemisphere_sampling(const float* n, uint32_t* state, float* out);
void orc_diffuse_sample(const float* n, uint32_t* state, float* out, float* pdf);
void orc_microfacet_sample(const float* n, const float* v, float a2, uint32_t* state, float* out, float* pdf);

/* diagnostic: count visits per inner node into per_node_counts[bvh_node_count] (NULL = off) */
void orc_set_visit_histogram(uint64_t* per_node_counts);
/* diagnostic, one thread only: trips (inner visits + leaves entered) of every traversal, bit 31 = any-hit walk, bit 30 = it ended on an accepted triangle */
void orc_set_trip_log(uint32_t* log, size_t capacity);
void orc_set_trip_log_tagged(uint32_t* log, size_t capacity); /* + a second word per traversal: pixel index << 4 | light steps done */
size_t orc_trip_log_length(void);
void orc_popcull_counts(uint64_t* out4, int reset_and_enable); /* experiment counters, see pt_oracle.c; -1: read only */
void orc_set_anyhit_order(int order); /* experiment: entry order of any-hit walks, see pt_oracle.c */

uint32_t orc_triple32(uint32_t x);
void orc_sincos(float x, float* s, float* c);
float orc_pow(float x, float y);
float orc_exp2(float x);
float orc_log2(float x);

#ifdef __cplusplus
}
#endif
#endif
