/*
 * pathtracer_amd.h — C ABI of the MI355X-native path-tracing hot path.
 *
 * This is the drop-in boundary that replaces src/renderer of Dolfun/pathtracer: the
 * Vulkan device bring-up, the descriptor/pipeline plumbing and the single compute
 * dispatch of src/renderer/shaders/main.comp.  Every entry point names the reference
 * interface it stands in for (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary: every call returns PT_OK (0) or a
 *     negative pt_status; pt_last_error() gives the text for the last failure
 *     (the reference throws std::runtime_error / vk::SystemError and main() prints
 *     them, src/main.cc:86-90).
 *   - all inputs are BORROWED for the duration of the call and never mutated; all
 *     outputs are written into CALLER-OWNED buffers (the reference returns a pointer
 *     into memory of an already destroyed RenderJob, src/renderer/renderer.cc:186-193).
 *   - every struct below is byte-for-byte the layout the reference host hands to its
 *     shader, so reference-packed data is accepted verbatim.
 *   - a pt_ctx is bound to one GPU and is not thread-safe (the reference is single
 *     threaded with one queue, src/renderer/render_job.cc:819-834).
 */
#ifndef PATHTRACER_AMD_H
#define PATHTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API __attribute__((visibility("default")))

typedef enum pt_status {
  PT_OK = 0,
  PT_ERR_INVALID = -1,   /* bad argument / inconsistent scene */
  PT_ERR_DEVICE = -2,    /* HIP runtime error */
  PT_ERR_NO_SCENE = -3,  /* render before upload */
  PT_ERR_LIMIT = -4      /* a stated limit exceeded (stack depth, counts) */
} pt_status;

/* ------------------------------------------------------------------ scene.h PODs */

/* Scene::Vertex, src/scene.h:9-16 (60 bytes, no padding) */
typedef struct pt_vertex {
  float position[3];
  float normal[3];
  float tangent[3];
  float bitangent[3];
  float texcoord[2];
  int32_t material_index;
} pt_vertex;

/* Scene::Material, src/scene.h:20-32 == GLSL `Material`, main.comp:42-58 (64 bytes) */
typedef struct pt_material {
  float base_color_factor[4];
  int32_t base_color_texture_index;
  float metallic_factor;
  float roughness_factor;
  int32_t metallic_roughness_texture_index;
  float normal_scale;
  int32_t normal_texture_index;
  float occlusion_strength;
  int32_t occlusion_texture_index;
  float emissive_factor[3];
  int32_t emissive_texture_index;
} pt_material;

/* Scene::DirectionalLight, src/scene.h:57-61 (alignas(16) -> 32 bytes) */
typedef struct pt_directional_light {
  float color[3];
  float intensity;
  float direction[3];
  float _pad;
} pt_directional_light;

/* Scene::PointLight, src/scene.h:63-67 (32 bytes) */
typedef struct pt_point_light {
  float color[3];
  float intensity;
  float position[3];
  float _pad;
} pt_point_light;

/* Scene::SamplerFilter_t / SamplerWarp_t, src/scene.h:34-40 */
enum { PT_FILTER_NEAREST = 0, PT_FILTER_LINEAR = 1 };
enum { PT_WRAP_CLAMP_TO_EDGE = 0, PT_WRAP_MIRRORED_REPEAT = 1, PT_WRAP_REPEAT = 2 };

/* Scene::Sampler, src/scene.h:42-45 */
typedef struct pt_sampler {
  int32_t mag_filter, min_filter;
  int32_t wrap_s, wrap_t;
} pt_sampler;

/* Scene::Texture, src/scene.h:52-55 */
typedef struct pt_texture {
  int32_t sampler_index;
  int32_t image_index;
} pt_texture;

/* Scene::Image, src/scene.h:47-50 — always RGBA8 (gltf_loader.cc:280-282) */
typedef struct pt_image {
  uint32_t width, height;
  const uint8_t* rgba8; /* width*height*4 bytes, row-major, row 0 first */
} pt_image;

/* Scene::Camera, src/scene.h:69-73; `lookat` is a DIRECTION (render_job.cc:851) */
typedef struct pt_camera {
  float position[3];
  float lookat[3];
  float vertical_fov;
} pt_camera;

/* BVHNode, src/bvh/bvh.h:36-41 (36 bytes) */
typedef struct pt_bvh_node {
  float aabb_min[3];
  float aabb_max[3];
  uint32_t left_child_index;
  uint32_t begin_index;
  uint32_t triangle_count;
} pt_bvh_node;

/* ------------------------------------------------------- render_job.h device PODs */

/* PackedVertexData, src/renderer/render_job.h:7-13 == GLSL `Vertex`, main.comp:25-33 */
typedef struct pt_packed_vertex {
  float position[4];
  float normal_and_texcoord_u[4];
  float tangent_and_texcoord_v[4];
  float bitangent[3];
  int32_t material_index;
} pt_packed_vertex;

/* PackedBVHNode, src/renderer/render_job.h:15-20 == GLSL `BVHNode`, main.comp:35-40 */
typedef struct pt_packed_bvh_node {
  float aabb_min[3];
  uint32_t left_or_begin;
  float aabb_max[3];
  uint32_t triangle_count;
} pt_packed_bvh_node;

/* PushConstants, src/renderer/render_job.h:22-38 == main.comp:123-130 (96 bytes) */
typedef struct pt_push_constants {
  float camera_position[3];   float _p0;
  float pixel_delta_u[3];     float _p1;
  float pixel_delta_v[3];     float _p2;
  float corner_pixel_pos[3];  float _p3;
  uint32_t resolution_x, resolution_y;
  uint32_t seed;
  uint32_t sample_count;
  float bg_color[3];
  uint32_t emissive_triangle_count;
} pt_push_constants;

/* RenderConfig, src/renderer/renderer.h:16-20 */
typedef struct pt_render_config {
  uint32_t resolution_x, resolution_y;
  uint32_t sample_count;
  float bg_color[3];
} pt_render_config;

/*
 * Everything RenderJob binds for the dispatch (descriptor bindings 0-7 and the
 * specialization constants, src/renderer/render_job.cc:33-40, 462-521, 563-570).
 */
typedef struct pt_scene_desc {
  uint32_t triangle_count;
  const float* vertex_positions;              /* binding 0: vec4 x 3 x triangle_count        */
  const pt_packed_vertex* packed_vertex_data; /* binding 1: 3 x triangle_count               */
  const pt_packed_bvh_node* bvh_nodes;        /* binding 2                                   */
  uint32_t bvh_node_count;
  uint32_t bvh_max_depth;                     /* specialization constant 5                   */
  const uint32_t* emissive_triangles;         /* binding 3 (may be NULL when count is 0)     */
  uint32_t emissive_triangle_count;
  const pt_material* materials;               /* binding 4, specialization constant 2        */
  uint32_t material_count;
  const pt_directional_light* directional_lights; /* binding 5                               */
  uint32_t directional_light_count;           /* spec constant 3: array size - 1 (the trailing dummy excluded) */
  const pt_point_light* point_lights;         /* binding 6                                   */
  uint32_t point_light_count;                 /* spec constant 4: array size - 1             */
  const pt_texture* textures;                 /* binding 7: combined image samplers          */
  uint32_t texture_count;
  const pt_image* images;
  uint32_t image_count;
  const pt_sampler* samplers;
  uint32_t sampler_count;
} pt_scene_desc;

/* The push-constant block plus the build-only fields the reference has no place for. */
typedef struct pt_render_params {
  pt_push_constants pc; /* pc.seed is explicit (the reference draws it from std::random_device, render_job.cc:841-844) */
  uint32_t rank;        /* sample-split data parallelism: this GPU renders strata   */
  uint32_t num_ranks;   /* [rank*n*n/num_ranks, (rank+1)*n*n/num_ranks); 0 or 1 = all */
  uint32_t flags;       /* reserved, 0 */
} pt_render_params;

/* Per-kernel slice of the work counters (with PT_FLAG_TIME_KERNELS also its device time). */
typedef struct pt_kernel_stats {
  uint64_t launches;
  uint64_t rays;         /* traversals started by this kernel            */
  uint64_t node_visits;
  uint64_t tri_tests;
  uint64_t hits_shaded;
  uint64_t texture_taps;
  double   ms;           /* sum of the launches' durations (HIP events on the launch stream) */
} pt_kernel_stats;

/* Work counters of one render, SURVEY.md §8d definitions. */
typedef struct pt_stats {
  uint64_t paths;          /* trace() calls (camera samples)                               */
  uint64_t rays;           /* ray_scene_intersect calls: closest-hit + every shadow-walk step */
  uint64_t node_visits;    /* inner-node visits (two child boxes tested each)              */
  uint64_t tri_tests;      /* ray_triangle_intersect calls                                 */
  uint64_t hits_shaded;    /* get_interpolated_data calls (shaded hits + shadow-walk steps + area-light samples) */
  uint64_t texture_taps;   /* texture() calls with index != -1                             */
  uint64_t shaded_vertices;/* path vertices run through the BRDF/light-sampling block      */
  uint64_t nan_samples;    /* samples dropped by the NaN filter (main.comp:608-609)        */
  uint64_t iterations;     /* wavefront iterations launched                                */
  double   render_ms;      /* device time of the render loop (HIP events)                  */
  double   upload_ms;      /* last pt_upload_scene H2D time                                */
  double   download_ms;    /* last D2H of the outputs                                      */
  pt_kernel_stats extend;  /* closest hit of every path segment (k_extend, k_paths)          */
  pt_kernel_stats shade;   /* shading, light sampling, regeneration, compaction (k_shade, k_paths) */
  pt_kernel_stats shadow;  /* the shadow walks (run inside k_shade / k_paths, or k_shadow): their traversal work */
  pt_kernel_stats path_kernel; /* k_paths: launches and device time only (its work is counted in the three slices above) */
  /* the ray mix behind `rays` (rays = rays_closest + rays_shadow_walked + rays_shadow_shortcut) */
  uint64_t rays_closest;         /* closest-hit traversals of path segments (main.comp:516)                    */
  uint64_t rays_shadow_walked;   /* shadow-walk steps that traversed the tree (main.comp:306-332)              */
  uint64_t rays_shadow_shortcut; /* all-opaque scenes: area-light samples decided by the one-triangle test of
                                    the sampled emitter, without a tree walk (the reference traverses; the result
                                    is provably the same). 0 from the CPU oracle's reference mode. */
  double   reduce_ms;            /* device group (pt_create_multi): time of the RCCL sum-reduce of the accumulators */
  uint64_t scene_in_lds;         /* 1: the scene was small enough to be staged whole in every workgroup's LDS
                                    (its fetches are LDS reads, not HBM traffic: the render is VALU-bound)      */
  uint64_t merged_walk;          /* 1: the path kernel was k_paths_merged (scene in global memory, at most two light samples
                                    per vertex): a vertex's shadow walks ride with the next closest-hit walk; all traversal
                                    work is then counted in the `extend` slice                                   */
  uint64_t bvh_wide4;            /* 1: PT_BVH=wide4 was in effect (opt-in, NOT bit-exact): node_visits counts visits of
                                    64-byte records of up to four children with 8-bit boxes                       */
  uint64_t fast_arith;           /* 1: PT_ARITH=fast was in effect (opt-in, NOT bit-exact; scenes staged in LDS only: scenes read
                                    from global memory wait on memory and run the exact kernels whatever PT_ARITH says) */
} pt_stats;

#define PT_FLAG_TIME_KERNELS 1u /* bracket every kernel with HIP events (serialises; for profiling) */

typedef struct pt_ctx pt_ctx;

/* ------------------------------------------------------------------- device side */

/* list_devices(), src/renderer/renderer.h:55, renderer.cc:196-213 */
PT_API int pt_device_count(void);
PT_API int pt_device_name(int device, char* out, size_t out_size);

/* Renderer::Renderer(device_index), src/renderer/renderer.h:24; device < 0 picks GPU 0
 * (the reference's -1 picks the first discrete GPU, renderer.cc:58-70). */
PT_API pt_ctx* pt_create(int device);
/* One context over several GPUs of THIS process (no counterpart in the single-device reference):
 * the scene is replicated in every GPU's HBM by pt_upload_scene, pt_render gives rank i the strata
 * [i n^2/n_dev, (i+1) n^2/n_dev) of every pixel (one host thread per GPU), sum-reduces the float4 accumulators to the
 * first device with ONE ncclReduce over a communicator per GPU (ncclCommInitAll; RCCL is loaded with dlopen on the first
 * call) and resolves there. With more than one device every rank draws from its own stream family, so
 * the image is the same estimate, not the same bits, as a one-GPU render; with one device it is bit-identical to it.
 * The other entry points (pt_render_accum_device, ...) act on the first device only. */
PT_API pt_ctx* pt_create_multi(const int* device_ids, int n_devices);
PT_API void pt_destroy(pt_ctx* ctx);

/* Text of the last error on ctx (or of the last failed pt_create when ctx is NULL). */
PT_API const char* pt_last_error(const pt_ctx* ctx);

/* RenderJob ctor's staging half: create_buffers / stage_* / copy_scene_data_to_device,
 * src/renderer/render_job.cc:33-51, 689-758.  Copies the scene into HBM (re-laid-out
 * for the kernels); the caller's arrays are not referenced after return. */
PT_API int pt_upload_scene(pt_ctx* ctx, const pt_scene_desc* scene);

/* RenderJob::render() + dispatch_compute_shader + copy_result_to_host,
 * src/renderer/render_job.cc:760-834: runs main.comp's work for every pixel and
 * returns the rgba8 image (resolution_x*resolution_y*4 bytes) in rgba8_out.
 * Optional build-only outputs (NULL to skip): accum_out = 4 floats per pixel
 * (sum R, sum G, sum B over the valid samples, valid-sample count);
 * primary_id_out = triangle index hit by the pixel's first camera ray
 * (0xFFFFFFFF on a miss); stats_out. */
PT_API int pt_render(pt_ctx* ctx, const pt_render_params* params,
                     uint8_t* rgba8_out, float* accum_out,
                     uint32_t* primary_id_out, pt_stats* stats_out);

/* Multi-GPU halves of pt_render (no counterpart in the reference, which is single
 * device): render this rank's strata into a DEVICE accumulator (float4 per pixel,
 * overwritten), to be sum-reduced over ranks by the caller (RCCL), then resolved.
 * `stream` is a hipStream_t. NULL is NOT the legacy default stream: it selects the
 * context's own stream, which is created non-blocking and is therefore unordered
 * against stream 0 and against a collective's stream. A caller that puts a collective
 * between the two calls must pass the (non-NULL) stream the collective is ordered on
 * to both, or synchronise in between. pt_render_accum_device returns after the render
 * loop has drained on `stream` (it polls the live-slot count); pt_resolve_device only
 * enqueues. */
PT_API int pt_render_accum_device(pt_ctx* ctx, const pt_render_params* params,
                                  float* accum_dev, uint32_t* primary_id_dev,
                                  void* stream, pt_stats* stats_out);
/* pt_render_accum_device for callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc, processes whose memory pool is handed out at
 * start-up). NO COUNTERPART IN THE REFERENCE. Like pt_render_accum_device the call is
 * HOST-BLOCKING (it polls the live-slot count and ends with a stream synchronisation), so
 * it must not be made on a stream that is being captured into a HIP graph. Same arguments, same results bit for bit, but the frame's working set
 * (per-slot path state, slot queues, queue counters, work-counter table) lives in a fixed
 * file-scope device array of the library (~26 MiB per GPU, held until the process ends)
 * instead of hipMalloc'd buffers, and nothing is allocated or freed on the way (no
 * hipMalloc / hipFree / hipEventCreate; the scene must already be uploaded).
 * Limit: the frame may have at most PT_NOALLOC_MAX_SLOTS pixel slots, a slot being one
 * pixel of the 8x8-tile grid, i.e. ceil(W/8)*ceil(H/8)*64 <= 262144 (512x512, BASELINE
 * configs[0]); larger frames fail with PT_ERR_LIMIT. PT_FLAG_TIME_KERNELS is rejected
 * (PT_ERR_INVALID). accum_dev is required; primary_id_dev and stats_out may be NULL.
 * Every call on a GPU, from any context, shares that GPU's one array, which keeps the
 * previous call's contents: each call re-initialises what must start at zero, and calls
 * that overlap on one GPU (two streams, two contexts, two threads) race. */
#define PT_NOALLOC_MAX_SLOTS 262144u
PT_API int pt_render_accum_device_noalloc(pt_ctx* ctx, const pt_render_params* params,
                                          float* accum_dev, uint32_t* primary_id_dev,
                                          void* stream, pt_stats* stats_out);
/* main.comp:616-619: divide by the valid-sample count, gamma, rgba8 store. */
PT_API int pt_resolve_device(pt_ctx* ctx, const float* accum_dev, uint32_t pixel_count,
                             uint8_t* rgba8_dev, void* stream);

/* --------------------------------------------------------------------- host side */

/* build_bvh(Scene&, bin_count, depth&), src/bvh/bvh.h:42, bvh.cc:161-169.
 * `triangles` (3 x uint32 per triangle) is permuted IN PLACE exactly as the reference
 * permutes scene.triangles (bvh.cc:66).  nodes_out must hold 2*triangle_count entries;
 * *node_count_out receives the used count. Bit-exact restatement, built with
 * -ffp-contract=off. */
PT_API int pt_build_bvh(const pt_vertex* vertices, uint32_t vertex_count,
                        uint32_t* triangles, uint32_t triangle_count,
                        uint32_t bin_count,
                        pt_bvh_node* nodes_out, uint32_t* node_count_out,
                        uint32_t* depth_out);
/* The same arrays (nodes, numbering, triangle permutation, depth: bit-identical to
 * pt_build_bvh with one thread) built by `threads` host threads; SURVEY.md §8f-4, no
 * counterpart in the reference, whose builder (bvh.cc:36-87) is a sequential recursion.
 * threads = 0: the machine's hardware threads, at most 16, or $PT_BUILD_THREADS;
 * threads = 1: the sequential builder. pt_build_bvh itself takes this route from
 * 65 536 triangles on. Vertex positions must be finite (with NaNs the sequential
 * builder's result depends on comparison order, which chunked reduction does not keep). */
PT_API int pt_build_bvh_mt(const pt_vertex* vertices, uint32_t vertex_count,
                           uint32_t* triangles, uint32_t triangle_count,
                           uint32_t bin_count,
                           pt_bvh_node* nodes_out, uint32_t* node_count_out,
                           uint32_t* depth_out, uint32_t threads);

/* RenderJob::process_scene(), src/renderer/render_job.cc:76-121.  Output capacities:
 * vertex_positions_out 12 floats/triangle, packed_vertex_out 3/triangle,
 * emissive_out triangle_count entries, packed_nodes_out node_count entries. */
PT_API int pt_process_scene(const pt_vertex* vertices, uint32_t vertex_count,
                            const uint32_t* triangles, uint32_t triangle_count,
                            const pt_material* materials, uint32_t material_count,
                            const pt_bvh_node* nodes, uint32_t node_count,
                            float* vertex_positions_out,
                            pt_packed_vertex* packed_vertex_out,
                            uint32_t* emissive_out, uint32_t* emissive_count_out,
                            pt_packed_bvh_node* packed_nodes_out);

/* PushConstants::PushConstants(config, scene), src/renderer/render_job.cc:836-873,
 * with the seed passed in instead of drawn from std::random_device. */
PT_API int pt_make_push_constants(const pt_render_config* config, const pt_camera* camera,
                                  uint32_t emissive_triangle_count, uint32_t seed,
                                  pt_push_constants* out);

/* RenderJob::initialize_work_group_size(), src/renderer/render_job.cc:63-74 — kept only
 * so a baseline run can state the reference's launch shape; the HIP path does not use it. */
PT_API void pt_reference_work_group_size(uint32_t resolution_x, uint32_t resolution_y,
                                         uint32_t* local_x, uint32_t* local_y);

/* The records PT_BVH=wide4 walks (opt-in; NO COUNTERPART IN THE REFERENCE: a layout derived from its
 * PackedBVHNode array), built on the host exactly as pt_upload_scene builds them — exported so that the
 * construction can be checked without a GPU. `nodes` must be a valid tree (children of an inner node at left_or_begin,
 * left_or_begin + 1; pt_upload_scene validates, this call does not). Each record is 64 bytes:
 * {origin.xyz, ex | ey << 8 | ez << 16} {entry 0..3 = ref << 7 | cnt, 0xFFFFFFFF = empty} {lo.x[4], lo.y[4], lo.z[4], hi.x[4]}
 * {hi.y[4], hi.z[4], 0, 0}; child c's box is origin + byte * 2^(e - 127) per axis, rounded outwards. records_out may be NULL to
 * query the count (at most node_count / 2); *stack_need_out = the most children a walk can have deferred at once. */
PT_API int pt_build_wide4(const pt_packed_bvh_node* nodes, uint32_t node_count, void* records_out, uint32_t capacity,
                          uint32_t* record_count_out, uint32_t* stack_need_out);

/* ------------------------------------------------------------------- front end */

/* load_gltf(path) -> Scene, src/gltf_loader/gltf_loader.h:5 (gltf_loader.cc:487-494): the
 * loaded scene as borrowed arrays in the scene.h layouts (valid until pt_host_scene_free).
 * Light arrays include the trailing dummy entry the loader appends (gltf_loader.cc:178-179). */
typedef struct pt_host_scene pt_host_scene;
typedef struct pt_host_scene_arrays {
  const pt_vertex* vertices;                      uint32_t vertex_count;
  const uint32_t* triangles;                      uint32_t triangle_count;   /* 3 indices each */
  const pt_material* materials;                   uint32_t material_count;
  const pt_image* images;                         uint32_t image_count;
  const pt_sampler* samplers;                     uint32_t sampler_count;
  const pt_texture* textures;                     uint32_t texture_count;
  const pt_directional_light* directional_lights; uint32_t directional_light_array_size;
  const pt_point_light* point_lights;             uint32_t point_light_array_size;
  pt_camera camera;
} pt_host_scene_arrays;

PT_API pt_host_scene* pt_host_scene_load(const char* gltf_path); /* NULL on failure: pt_host_last_error() */
PT_API int pt_host_scene_view(const pt_host_scene* scene, pt_host_scene_arrays* out);
PT_API void pt_host_scene_free(pt_host_scene* scene);
PT_API const char* pt_host_last_error(void);

/* stbi_write_png(path, w, h, 4, data, stride), src/main.cc:78-83 */
PT_API int pt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8, uint32_t stride_bytes);
/* stb_image decode as tinygltf configures it (4 x 8-bit): PNG and baseline JPEG in this build. rgba8_out
 * may be NULL to query the size. */
PT_API int pt_decode_image(const uint8_t* bytes, size_t size, uint32_t* width, uint32_t* height,
                           uint8_t* rgba8_out, size_t out_capacity);

#ifdef __cplusplus
}
#endif

#endif /* PATHTRACER_AMD_H */
