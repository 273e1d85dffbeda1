// C-ABI device entry points: context, scene upload, the wavefront render loop.
// Replaces src/renderer/renderer.cc (device bring-up) and the Vulkan half of
// src/renderer/render_job.cc (buffers, staging copies, dispatch, copy-back) with
// hipMalloc / hipMemcpyAsync / kernel launches on one HIP stream.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the functions are resolved with dlsym when pt_create_multi first needs them

#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pathtracer_amd.h"
#include "pt_device.h"
#include "pt_glsl_math.h"

namespace {

thread_local std::string g_create_error;

// Slots per band (see render_accum). 0 = no banding. Tunable through PT_BAND_SLOTS for experiments.
constexpr uint32_t kDefaultBandSlots = 0;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// ---- scratch of pt_render_accum_device_noalloc: the per-slot path state, queues, queue counters and work-counter
// table of ONE frame of at most PT_NOALLOC_MAX_SLOTS pixel slots, in a file-scope device array instead of hipMalloc'd
// buffers. Each GPU the library runs on holds its own copy (at its own address) until the process ends. In host code
// the NAME `scratch` is only a handle for hipGetSymbolAddress: the address every launch and memset uses is the
// per-device pointer pt_create() looks up once, with that device current (g_scratch_ptr).
constexpr uint32_t kNoAllocSlots = PT_NOALLOC_MAX_SLOTS;
constexpr uint32_t kNoAllocStatRows = kNoAllocSlots / 64 * 2 + 16;
constexpr size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
constexpr size_t kOffRayO = 0;
constexpr size_t kOffRayD = kOffRayO + align256(16ull * kNoAllocSlots);
constexpr size_t kOffPathA = kOffRayD + align256(16ull * kNoAllocSlots);
constexpr size_t kOffPathB = kOffPathA + align256(16ull * kNoAllocSlots);
constexpr size_t kOffAccum = kOffPathB + align256(16ull * kNoAllocSlots);
constexpr size_t kOffPrimary = kOffAccum + align256(16ull * kNoAllocSlots);
constexpr size_t kOffQueue0 = kOffPrimary + align256(4ull * kNoAllocSlots);
constexpr size_t kOffQueue1 = kOffQueue0 + align256(4ull * kNoAllocSlots);
constexpr size_t kOffCounters = kOffQueue1 + align256(4ull * kNoAllocSlots);
constexpr size_t kOffStats = kOffCounters + align256(PT_CNT_WORDS * sizeof(uint32_t));
constexpr size_t kOffTotals = kOffStats + align256(sizeof(unsigned long long) * PT_K_COUNT * kNoAllocStatRows * PT_ST_WORDS);
constexpr size_t kScratchBytes = kOffTotals + align256(sizeof(unsigned long long) * PT_K_COUNT * PT_ST_WORDS);
static_assert(kScratchBytes <= (32ull << 20), "scratch is meant to stay modest");

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};  // per device: device address of that GPU's copy of `scratch` (under g_scratch_mutex)
std::mutex g_scratch_mutex;                      // pt_create may run on several host threads at once

// RCCL, loaded on demand (pt_create_multi): one communicator per GPU of a single-process device group, used for exactly one
// collective per frame — the sum-reduce of the float4 accumulators to the group's first GPU. No link-time
// dependency: single-GPU users and the one-process-per-GPU path (torch.distributed) never load it through this library.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.lib || !api.error.empty()) return api;
  // PT_RCCL_LIB names the library to load instead (a site's own build; the CPU suite points it at a missing file to see the error path)
  const char* override_name = std::getenv("PT_RCCL_LIB");
  std::string last_error;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(override_name ? override_name : name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
    const char* e = dlerror();  // returns the message ONCE and clears it
    last_error = e ? e : "dlopen failed";
    if (override_name) break;
  }
  if (!api.lib) {
    api.error = std::string("librccl.so not found: ") + last_error;
    return api;
  }
  auto sym = [&](const char* n) {
    void* p = dlsym(api.lib, n);
    if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + n;
    return p;
  };
  api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Reduce = reinterpret_cast<decltype(api.Reduce)>(sym("ncclReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  return api;
}

}  // namespace

struct pt_ctx {
  // device group (pt_create_multi): this context is rank 0; `members` lists every rank's context (this one first),
  // `comms` their RCCL communicators. Empty for a plain pt_create context.
  std::vector<pt_ctx*> members;
  std::vector<ncclComm_t> comms;
  double reduce_ms = 0.0;

  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // scene
  bool has_scene = false;
  pt_dev_scene scene{};
  uint32_t emissive_count = 0;
  std::vector<DevBuf> scene_bufs;
  double upload_ms = 0.0;

  // per-resolution working set
  pt_dev_state state{};
  uint32_t state_slots = 0;
  uint32_t state_lights = 0;
  std::vector<DevBuf> state_bufs;
  float4* accum_out = nullptr;     // pixel order
  uint32_t* primary_out = nullptr;
  uint32_t* rgba8_out = nullptr;
  uint32_t out_pixels = 0;

  uint32_t band_slots = 0;          // 0 = whole frame in one band
  uint32_t trace_blocks = 0;        // grid cap of k_extend (persistent workgroups striding the queue)
  uint32_t shade_blocks = 0;        // grid cap of k_shade
  bool log_launches = false;        // PT_LOG_LAUNCHES: print the polled live-slot count of every launch to stderr
  bool fast_arith = false;          // PT_ARITH=fast: k_paths of the fast-arithmetic build (path pipeline only; opt-in, not bit-exact)
  int pipeline = 0;                 // 0: path kernel (k_paths); 1: wavefront (k_extend + k_shade [+ k_shadow]); PT_PIPELINE=paths|wavefront
  uint32_t paths_vertices = 32;     // path vertices a slot advances per k_paths launch (PT_PATHS_VERTICES)
  uint32_t paths_blocks = 0xFFFFFFFFu; // grid cap of k_paths (PT_PATHS_BLOCKS)
  uint32_t paths_block_threads = 64;   // workgroup size of k_paths (PT_PATHS_BLOCK_THREADS: 64, 128 or 256)
  uint32_t xcd_chunk = 0;           // k_paths: deal the queue to the XCDs in chunks of this many workgroups (PT_XCD_CHUNK; 0 = plain round-robin)
  uint32_t compact_every = 16;      // fused pipeline: iterations between queue compactions (PT_COMPACT_EVERY; 1 = every iteration)
  bool split_shadow = true;         // scenes in global memory: shadow walks in their own kernel (PT_SPLIT_SHADOW=0 keeps them in k_shade)
  uint32_t tail_entries = 262144;   // queue length below which the rest of the frame runs in one k_tail launch (0 = never)
  uint32_t lds_scene_bytes = 16384; // scenes up to this size (pair records + triangle positions) are staged whole in LDS
  uint32_t lds_stack = 16;          // PT_LDS_STACK: traversal-stack rows in LDS per lane for scenes in global memory (0 = the whole depth).
                                    // 16 rows + private-memory overflow: the 4.7 M-triangle rocks (depth 33) go from 4 to 7 k_extend waves
                                    // per SIMD (-21 % k_extend time) and k_paths +5 %; shallower trees are unaffected (measured)
  bool wide4 = false;               // PT_BVH=wide4 (opt-in, NOT bit-exact): k_paths_merged walks 64-byte records of up to four children with 8-bit boxes
  bool merged = true;               // PT_MERGED=0: scenes in global memory with at most two light samples per vertex run k_paths (shadow walks in
                                    // their own phase) instead of k_paths_merged (a vertex's shadow walks ride with the next closest-hit walk)
  uint32_t pair_order = 0;          // PT_PAIR_ORDER: order of the child-pair records in memory (pt_upload_scene; same results for any value)
  uint32_t pair_order_levels = 8;   // PT_PAIR_ORDER_LEVELS: levels kept breadth-first by PT_PAIR_ORDER=2
  uint32_t merged_lds_stack = 0;    // 1 + PT_MERGED_LDS_STACK, 0: unset = 9 rows with two light samples per vertex, 16 with one (pt_render). LDS stack rows per
                                    // lane of k_paths_merged, deeper rows in private memory. With two samples 36 B + 80 B of stash per
                                    // lane keep 20 waves per CU inside the 160 KB (measured, rocks / atrium Mrays/s: 4 rows 1461 / 1961, 6: 1529 / 2025,
                                    // 8: 1601 / 2038, 9: 1623 / 2040, 10: 1544 / 1933, 11: 1549 / 1933, 16: fewer waves fit)
  bool child_entries = true;        // PT_CHILD_ENTRIES=0: traversal-stack entries as (record, side) even where the scene allows (ref << 7 | cnt)
  uint32_t* host_counts = nullptr;  // pinned: [0..1] polled live-slot counts, from byte 64 on the work-counter totals
  unsigned char* scratch = nullptr; // this device's copy of the file-scope scratch array (pt_render_accum_device_noalloc)
  pt_dev_state scratch_state{};     // the scratch carved into the per-slot buffers of a frame of kNoAllocSlots slots
  hipEvent_t poll_events[2] = {nullptr, nullptr};
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  std::vector<hipEvent_t> kernel_events;

  // Concurrent bands (path pipeline, PT_CONCURRENT_BANDS): every band of the frame has its own stream, poll events and pinned
  // words, and bands 1.. their own per-slot state and queues (band 0 uses `state`). The streams are the library's own, created
  // one after another, so that they land on different hardware queues (HIP deals streams to GPU_MAX_HW_QUEUES = 4 queues in
  // creation order; a band on the caller's stream shared a queue with another band and the frame got slower, not faster);
  // the caller's stream only forks and joins them.
  struct Band {
    pt_dev_state state{};
    std::vector<DevBuf> bufs;
    uint32_t slots = 0;
    hipStream_t stream = nullptr;
    hipEvent_t poll_events[2] = {nullptr, nullptr};
    hipEvent_t done = nullptr;
    uint32_t* host_counts = nullptr;
  };
  std::vector<Band> bands;
  uint32_t concurrent_bands = 2;    // measured on cfg2: 2 bands -10 % frame time wherever tried; 3 and 4 the same when every band's stream
                                    // has a hardware queue to itself and SLOWER than one band when two of them share one (which
                                    // depends on what else the process has created: 4 bands gave 91 / 95 / 105 ms in three hosts); 6, 8 worse
  uint32_t band_min_tiles = 4096;   // PT_BAND_MIN_TILES: fewer concurrent bands rather than bands below this many 8x8 tiles
  hipEvent_t fork_event = nullptr;
};

namespace {

int fail(pt_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_create_error = buf;
  return code;
}

#define PT_HIP(ctx, expr)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail((ctx), PT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                         \
  } while (0)

void free_bufs(std::vector<DevBuf>& v) {
  for (auto& b : v)
    if (b.p) (void)hipFree(b.p);
  v.clear();
}

template <class T>
int dev_alloc(pt_ctx* ctx, std::vector<DevBuf>& owner, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  PT_HIP(ctx, hipMalloc(&p, bytes));
  owner.push_back({p, bytes});
  *out = static_cast<T*>(p);
  return PT_OK;
}

template <class T>
int dev_upload(pt_ctx* ctx, std::vector<DevBuf>& owner, const T** out, const T* host, size_t count) {
  T* d = nullptr;
  int rc = dev_alloc(ctx, owner, &d, count);
  if (rc != PT_OK) return rc;
  if (count) PT_HIP(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  *out = d;
  return PT_OK;
}

// Host-side shape checks: every index the kernels will follow is in range before anything
// is launched. Returns the real tree depth (root = 1) through depth_out.
int validate_scene(pt_ctx* ctx, const pt_scene_desc* s, uint32_t* depth_out) {
  if (!s->triangle_count || !s->vertex_positions || !s->packed_vertex_data || !s->bvh_nodes || !s->materials)
    return fail(ctx, PT_ERR_INVALID, "scene: missing triangle / node / material arrays");
  if (s->bvh_node_count < 1 || s->material_count < 1)
    return fail(ctx, PT_ERR_INVALID, "scene: empty node or material array");
  if (s->emissive_triangle_count && !s->emissive_triangles)
    return fail(ctx, PT_ERR_INVALID, "scene: emissive list missing");
  if ((s->directional_light_count && !s->directional_lights) || (s->point_light_count && !s->point_lights))
    return fail(ctx, PT_ERR_INVALID, "scene: light array missing");

  // tree walk: bounds of every reference + depth
  std::vector<std::pair<uint32_t, uint32_t>> todo;  // node, depth
  todo.emplace_back(0u, 1u);
  uint32_t max_depth = 0;
  size_t visited = 0;
  while (!todo.empty()) {
    auto [ni, depth] = todo.back();
    todo.pop_back();
    if (++visited > s->bvh_node_count) return fail(ctx, PT_ERR_INVALID, "scene: BVH is not a tree");
    if (depth > max_depth) max_depth = depth;
    const pt_packed_bvh_node& n = s->bvh_nodes[ni];
    if (n.triangle_count > 0) {
      if (static_cast<uint64_t>(n.left_or_begin) + n.triangle_count > s->triangle_count)
        return fail(ctx, PT_ERR_INVALID, "scene: leaf %u references triangles out of range", ni);
    } else {
      if ((n.left_or_begin & 1u) || n.left_or_begin < 2 || n.left_or_begin + 1 >= s->bvh_node_count)
        return fail(ctx, PT_ERR_INVALID, "scene: node %u has a bad child pair %u", ni, n.left_or_begin);
      todo.emplace_back(n.left_or_begin + 1, depth + 1);
      todo.emplace_back(n.left_or_begin, depth + 1);
    }
  }
  *depth_out = max_depth;

  for (uint32_t t = 0; t < s->triangle_count; ++t) {
    int32_t mi = s->packed_vertex_data[3 * static_cast<size_t>(t)].material_index;
    if (mi < 0 || static_cast<uint32_t>(mi) >= s->material_count)
      return fail(ctx, PT_ERR_INVALID, "scene: triangle %u has material index %d out of range", t, mi);
  }
  for (uint32_t i = 0; i < s->emissive_triangle_count; ++i)
    if (s->emissive_triangles[i] >= s->triangle_count)
      return fail(ctx, PT_ERR_INVALID, "scene: emissive triangle index out of range");
  auto tex_ok = [&](int32_t idx) { return idx == -1 || (idx >= 0 && static_cast<uint32_t>(idx) < s->texture_count); };
  for (uint32_t i = 0; i < s->material_count; ++i) {
    const pt_material& m = s->materials[i];
    if (!tex_ok(m.base_color_texture_index) || !tex_ok(m.metallic_roughness_texture_index) ||
        !tex_ok(m.occlusion_texture_index) || !tex_ok(m.emissive_texture_index))
      return fail(ctx, PT_ERR_INVALID, "scene: material %u has a texture index out of range", i);
  }
  for (uint32_t i = 0; i < s->texture_count; ++i) {
    const pt_texture& t = s->textures[i];
    if (t.sampler_index < 0 || static_cast<uint32_t>(t.sampler_index) >= s->sampler_count || t.image_index < 0 ||
        static_cast<uint32_t>(t.image_index) >= s->image_count)
      return fail(ctx, PT_ERR_INVALID, "scene: texture %u references a sampler/image out of range", i);
  }
  for (uint32_t i = 0; i < s->image_count; ++i) {
    const pt_image& im = s->images[i];
    if (!im.width || !im.height || !im.rgba8 || im.width > 32768 || im.height > 32768)
      return fail(ctx, PT_ERR_INVALID, "scene: image %u is empty or too large", i);
  }
  return PT_OK;
}

// A frame rendered as n concurrent bands: a stream, two poll events, a join event and pinned words per band, and per-slot
// state of `slots` slots (path pipeline: no shadow requests) for bands 1..n-1; created on first use and kept.
int ensure_bands(pt_ctx* ctx, uint32_t n, uint32_t slots) {
  if (ctx->bands.size() < n) ctx->bands.resize(n);
  if (!ctx->fork_event) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
  for (uint32_t b = 0; b < n; ++b)  // the streams first, back to back
    if (!ctx->bands[b].stream) PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->bands[b].stream, hipStreamNonBlocking));
  for (uint32_t b = 0; b < n; ++b) {
    pt_ctx::Band& band = ctx->bands[b];
    if (!band.done) {
      PT_HIP(ctx, hipEventCreate(&band.poll_events[0]));
      PT_HIP(ctx, hipEventCreate(&band.poll_events[1]));
      PT_HIP(ctx, hipEventCreateWithFlags(&band.done, hipEventDisableTiming));
      PT_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&band.host_counts), 64 + sizeof(unsigned long long) * PT_K_COUNT * PT_ST_WORDS,
                                hipHostMallocDefault));
    }
    if (b == 0 || band.slots == slots) continue;
    free_bufs(band.bufs);
    band.slots = 0;
    pt_dev_state& st = band.state;
    st = pt_dev_state{};
    int rc;
#define ALLOC(field, count) \
  if ((rc = dev_alloc(ctx, band.bufs, &field, (count))) != PT_OK) return rc
    ALLOC(st.ray_o, slots);
    ALLOC(st.ray_d, slots);
    ALLOC(st.path_a, slots);
    ALLOC(st.path_b, slots);
    ALLOC(st.accum, slots);
    ALLOC(st.primary_id, slots);
    ALLOC(st.queue[0], slots);
    ALLOC(st.queue[1], slots);
    st.slot_stride = slots;
    ALLOC(st.counters, PT_CNT_WORDS);
    st.stat_rows = slots / 64 * 2 + 16;
    ALLOC(st.stats, static_cast<size_t>(PT_K_COUNT) * st.stat_rows * PT_ST_WORDS);
    ALLOC(st.stat_totals, static_cast<size_t>(PT_K_COUNT) * PT_ST_WORDS);
#undef ALLOC
    band.slots = slots;
  }
  return PT_OK;
}

int ensure_state(pt_ctx* ctx, uint32_t slots, uint32_t lights, uint32_t pixels) {
  if (ctx->state_slots == slots && ctx->state_lights == lights && ctx->out_pixels == pixels) return PT_OK;
  free_bufs(ctx->state_bufs);
  ctx->state_slots = 0;
  pt_dev_state& st = ctx->state;
  int rc;
#define ALLOC(field, count) \
  if ((rc = dev_alloc(ctx, ctx->state_bufs, &field, (count))) != PT_OK) return rc
  ALLOC(st.ray_o, slots);
  ALLOC(st.ray_d, slots);
  ALLOC(st.hit, slots);
  ALLOC(st.path_a, slots);
  ALLOC(st.path_b, slots);
  ALLOC(st.accum, slots);
  ALLOC(st.primary_id, slots);
  ALLOC(st.queue[0], slots);
  ALLOC(st.queue[1], slots);
  st.slot_stride = slots;
  // split pipeline buffers (only used when the scene is read from global memory and has lights)
  ALLOC(st.pend, lights ? 5 * static_cast<size_t>(slots) : 0);
  ALLOC(st.req, 4 * static_cast<size_t>(lights) * slots);
  ALLOC(st.queue_shadow, lights ? slots : 0);
  ALLOC(st.counters, PT_CNT_WORDS);
  st.stat_rows = slots / 64 * 2 + 16;  // >= the largest grid (64-thread blocks, rounded up to a multiple of 8)
  ALLOC(st.stats, static_cast<size_t>(PT_K_COUNT) * st.stat_rows * PT_ST_WORDS);
  ALLOC(st.stat_totals, static_cast<size_t>(PT_K_COUNT) * PT_ST_WORDS);
  ALLOC(ctx->accum_out, pixels);
  ALLOC(ctx->primary_out, pixels);
  ALLOC(ctx->rgba8_out, pixels);
#undef ALLOC
  ctx->state_slots = slots;
  ctx->state_lights = lights;
  ctx->out_pixels = pixels;
  return PT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI

extern "C" int pt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int pt_device_name(int device, char* out, size_t out_size) {
  if (!out || !out_size) return PT_ERR_INVALID;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PT_ERR_DEVICE;
  snprintf(out, out_size, "%s (%s, %d CUs, %.0f GiB)", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           static_cast<double>(prop.totalGlobalMem) / (1024.0 * 1024.0 * 1024.0));
  return PT_OK;
}

extern "C" pt_ctx* pt_create(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    fail(nullptr, PT_ERR_DEVICE, "no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "count is 0");
    return nullptr;
  }
  if (device < 0) device = 0;
  if (device >= n) {
    fail(nullptr, PT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    return nullptr;
  }
  pt_ctx* ctx = new pt_ctx();
  ctx->device = device;
  ctx->band_slots = kDefaultBandSlots;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      ctx->trace_blocks = 32u * static_cast<uint32_t>(prop.multiProcessorCount);  // measured on cfg2: 8192 > 4096 > 2048
      ctx->shade_blocks = 8u * static_cast<uint32_t>(prop.multiProcessorCount);   // measured on cfg2: 2048 > 4096 > 8192
    }
  }
  if (const char* e = std::getenv("PT_SHADE_BLOCKS")) ctx->shade_blocks = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->shade_blocks == 0) ctx->shade_blocks = 2048;
  if (const char* e = std::getenv("PT_TRACE_BLOCKS")) ctx->trace_blocks = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_LDS_SCENE_BYTES")) ctx->lds_scene_bytes = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->trace_blocks == 0) ctx->trace_blocks = 2048;
  if (ctx->lds_scene_bytes > 32768) ctx->lds_scene_bytes = 32768;
  ctx->log_launches = std::getenv("PT_LOG_LAUNCHES") != nullptr;
  if (const char* e = std::getenv("PT_ARITH")) ctx->fast_arith = std::string(e) == "fast";
  if (const char* e = std::getenv("PT_PIPELINE")) ctx->pipeline = (std::string(e) == "wavefront") ? 1 : 0;
  if (const char* e = std::getenv("PT_PATHS_VERTICES")) ctx->paths_vertices = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->paths_vertices == 0) ctx->paths_vertices = 1;
  if (const char* e = std::getenv("PT_PATHS_BLOCK_THREADS")) {
    const unsigned long v = std::strtoul(e, nullptr, 10);
    if (v == 64 || v == 128 || v == 256) ctx->paths_block_threads = static_cast<uint32_t>(v);
  }
  if (const char* e = std::getenv("PT_PATHS_BLOCKS")) ctx->paths_blocks = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->paths_blocks == 0) ctx->paths_blocks = 0xFFFFFFFFu;
  if (const char* e = std::getenv("PT_XCD_CHUNK")) ctx->xcd_chunk = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_COMPACT_EVERY")) ctx->compact_every = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_SPLIT_SHADOW")) ctx->split_shadow = std::strtoul(e, nullptr, 10) != 0;
  if (const char* e = std::getenv("PT_TAIL_ENTRIES")) ctx->tail_entries = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_LDS_STACK")) ctx->lds_stack = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->lds_stack > 64) ctx->lds_stack = 64;  // 256 threads x 64 rows x 4 B = the 64 KiB a workgroup may ask for
  if (const char* e = std::getenv("PT_CHILD_ENTRIES")) ctx->child_entries = std::atoi(e) != 0;
  if (const char* e = std::getenv("PT_MERGED")) ctx->merged = std::atoi(e) != 0;
  if (const char* e = std::getenv("PT_BVH")) ctx->wide4 = std::string(e) == "wide4";
  if (const char* e = std::getenv("PT_PAIR_ORDER")) ctx->pair_order = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->pair_order > 4) ctx->pair_order = 0;
  if (const char* e = std::getenv("PT_PAIR_ORDER_LEVELS")) ctx->pair_order_levels = std::min(30ul, std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_MERGED_LDS_STACK")) ctx->merged_lds_stack = 1u + std::min(64u, static_cast<uint32_t>(std::strtoul(e, nullptr, 10)));
  if (const char* e = std::getenv("PT_CONCURRENT_BANDS")) ctx->concurrent_bands = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (const char* e = std::getenv("PT_BAND_MIN_TILES")) ctx->band_min_tiles = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  if (ctx->band_min_tiles < 1) ctx->band_min_tiles = 1;
  if (ctx->concurrent_bands < 1) ctx->concurrent_bands = 1;
  if (ctx->concurrent_bands > 8) ctx->concurrent_bands = 8;
  if (const char* e = std::getenv("PT_BAND_SLOTS")) ctx->band_slots = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
  bool ok = hipSetDevice(device) == hipSuccess &&
            hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&ctx->host_counts), 64 + sizeof(unsigned long long) * PT_K_COUNT * PT_ST_WORDS,
                          hipHostMallocDefault) == hipSuccess &&
            hipEventCreate(&ctx->poll_events[0]) == hipSuccess && hipEventCreate(&ctx->poll_events[1]) == hipSuccess &&
            hipEventCreate(&ctx->ev_start) == hipSuccess && hipEventCreate(&ctx->ev_stop) == hipSuccess;
  if (!ok) {
    fail(nullptr, PT_ERR_DEVICE, "context set-up failed on device %d: %s", device, hipGetErrorString(hipGetLastError()));
    pt_destroy(ctx);
    return nullptr;
  }
  // The no-allocation entry point's scratch: once per device, with that device current (set above), look up where THIS
  // GPU holds its copy of the file-scope array. Only this pointer is ever handed to a kernel, memset or copy.
  if (device < kMaxDevices) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
      void* sym = nullptr;
      const hipError_t se = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
      if (se != hipSuccess || !sym) {
        fail(nullptr, PT_ERR_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(se));
        pt_destroy(ctx);
        return nullptr;
      }
      g_scratch_ptr[device] = static_cast<unsigned char*>(sym);
    }
    ctx->scratch = g_scratch_ptr[device];
    unsigned char* b = ctx->scratch;
    pt_dev_state& ss = ctx->scratch_state;
    ss.ray_o = reinterpret_cast<float4*>(b + kOffRayO);
    ss.ray_d = reinterpret_cast<float4*>(b + kOffRayD);
    ss.path_a = reinterpret_cast<float4*>(b + kOffPathA);
    ss.path_b = reinterpret_cast<float4*>(b + kOffPathB);
    ss.accum = reinterpret_cast<float4*>(b + kOffAccum);
    ss.primary_id = reinterpret_cast<uint32_t*>(b + kOffPrimary);
    ss.queue[0] = reinterpret_cast<uint32_t*>(b + kOffQueue0);
    ss.queue[1] = reinterpret_cast<uint32_t*>(b + kOffQueue1);
    ss.counters = reinterpret_cast<uint32_t*>(b + kOffCounters);
    ss.stats = reinterpret_cast<unsigned long long*>(b + kOffStats);
    ss.stat_totals = reinterpret_cast<unsigned long long*>(b + kOffTotals);
    ss.stat_rows = kNoAllocStatRows;
    ss.slot_stride = kNoAllocSlots;
    // hit / pend / req / queue_shadow stay NULL: only the wavefront pipeline reads them and this entry point never runs it
  }
  return ctx;
}

extern "C" void pt_destroy(pt_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->members.empty()) {  // a device group: communicators first, then the other ranks' contexts
    for (size_t i = 0; i < ctx->comms.size(); ++i)
      if (ctx->comms[i] && rccl().CommDestroy) (void)rccl().CommDestroy(ctx->comms[i]);
    ctx->comms.clear();
    std::vector<pt_ctx*> members;
    members.swap(ctx->members);
    for (pt_ctx* m : members)
      if (m != ctx) pt_destroy(m);
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_bufs(ctx->scene_bufs);
  free_bufs(ctx->state_bufs);
  for (pt_ctx::Band& band : ctx->bands) {
    if (band.stream) (void)hipStreamSynchronize(band.stream);
    free_bufs(band.bufs);
    if (band.host_counts) (void)hipHostFree(band.host_counts);
    for (hipEvent_t e : band.poll_events)
      if (e) (void)hipEventDestroy(e);
    if (band.done) (void)hipEventDestroy(band.done);
    if (band.stream) (void)hipStreamDestroy(band.stream);
  }
  if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
  if (ctx->host_counts) (void)hipHostFree(ctx->host_counts);
  for (hipEvent_t e : ctx->poll_events)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->kernel_events) (void)hipEventDestroy(e);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int upload_one(pt_ctx* ctx, const pt_scene_desc* s);

extern "C" int pt_upload_scene(pt_ctx* ctx, const pt_scene_desc* s) {
  if (!ctx) return PT_ERR_INVALID;
  if (!s) return fail(ctx, PT_ERR_INVALID, "scene is NULL");
  if (ctx->members.empty()) return upload_one(ctx, s);
  for (pt_ctx* m : ctx->members) {  // device group: the scene is replicated in every GPU's HBM
    const int rc = upload_one(m, s);
    if (rc != PT_OK) {
      if (m != ctx) ctx->err = "device " + std::to_string(m->device) + ": " + m->err;
      return rc;
    }
  }
  return PT_OK;
}

namespace {

// PT_BVH=wide4: the reference's binary tree collapsed into records of up to four children (pt_kernels.hip, wide_visit). A record
// starts from a binary inner node's two children and replaces the inner child with the largest box by its own two children until
// four are held or only leaves are left. Child boxes are stored as bytes on a per-record grid (common origin = the union's lower
// corner, per-axis power-of-two cell), rounded OUTWARDS and checked with the float arithmetic a decoder would use, so every stored
// box contains the reference's. Entries are (ref << 7 | cnt) like the kernels' stack entries (the caller checks packed_stack).
struct WideRecord {
  float origin[3];
  uint32_t exps;
  uint32_t entry[4];
  uint32_t lo[3], hi_x;
  uint32_t hi_y, hi_z, pad[2];
};
static_assert(sizeof(WideRecord) == 64, "layout");

void build_wide(const pt_scene_desc* s, std::vector<WideRecord>& rec, uint32_t& stack_need) {
  const pt_packed_bvh_node* N = s->bvh_nodes;
  rec.clear();
  stack_need = 0;
  if (N[0].triangle_count != 0) return;  // a single leaf: nothing to collapse
  auto area = [&](uint32_t n) {
    const float dx = N[n].aabb_max[0] - N[n].aabb_min[0], dy = N[n].aabb_max[1] - N[n].aabb_min[1], dz = N[n].aabb_max[2] - N[n].aabb_min[2];
    return dx * dy + dy * dz + dz * dx;
  };
  std::vector<uint32_t> order{0u};            // the binary inner node each wide record starts from, breadth-first
  std::vector<std::array<uint32_t, 4>> kid_wide;  // per record: wide index of an inner child, 0xFFFFFFFF for a leaf or an empty place
  std::vector<uint32_t> kid_count;
  for (size_t head = 0; head < order.size(); ++head) {
    uint32_t kids[4] = {N[order[head]].left_or_begin, N[order[head]].left_or_begin + 1u, 0u, 0u};
    uint32_t nk = 2;
    while (nk < 4) {
      int best = -1;
      for (uint32_t k = 0; k < nk; ++k)
        if (N[kids[k]].triangle_count == 0 && (best < 0 || area(kids[k]) > area(kids[best]))) best = static_cast<int>(k);
      if (best < 0) break;
      const uint32_t c = N[kids[best]].left_or_begin;
      kids[best] = c;
      kids[nk++] = c + 1u;
    }
    WideRecord r{};
    float lo_u[3], hi_u[3];
    for (int a = 0; a < 3; ++a) {
      lo_u[a] = N[kids[0]].aabb_min[a];
      hi_u[a] = N[kids[0]].aabb_max[a];
      for (uint32_t k = 1; k < nk; ++k) {
        lo_u[a] = std::fmin(lo_u[a], N[kids[k]].aabb_min[a]);
        hi_u[a] = std::fmax(hi_u[a], N[kids[k]].aabb_max[a]);
      }
      r.origin[a] = lo_u[a];
    }
    uint32_t q_lo[3][4] = {}, q_hi[3][4] = {};
    for (int a = 0; a < 3; ++a) {
      int e2 = 0;
      const float ext = hi_u[a] - lo_u[a];
      (void)std::frexp(ext > 0.0f ? ext / 255.0f : 0.0f, &e2);  // ext / 255 < 2^e2
      int E = ext > 0.0f ? e2 + 127 : 1;
      if (E < 1) E = 1;
      for (;; ++E) {  // the smallest cell that covers the extent with 255 steps, outward rounding included
        if (E > 254) E = 254;
        const float cell = std::ldexp(1.0f, E - 127);
        bool ok = true;
        for (uint32_t k = 0; k < nk && ok; ++k) {
          const float lo = N[kids[k]].aabb_min[a], hi = N[kids[k]].aabb_max[a];
          double ql = std::floor((static_cast<double>(lo) - lo_u[a]) / cell), qh = std::ceil((static_cast<double>(hi) - lo_u[a]) / cell);
          if (ql < 0.0) ql = 0.0;
          if (ql > 255.0) ql = 255.0;
          while (ql > 0.0 && lo_u[a] + static_cast<float>(ql) * cell > lo) ql -= 1.0;
          if (qh < 0.0) qh = 0.0;
          while (qh <= 255.0 && lo_u[a] + static_cast<float>(qh) * cell < hi) qh += 1.0;
          if (qh > 255.0) ok = false;
          q_lo[a][k] = static_cast<uint32_t>(ql);
          q_hi[a][k] = static_cast<uint32_t>(qh);
        }
        if (ok || E == 254) break;
      }
      r.exps |= static_cast<uint32_t>(E) << (8 * a);
    }
    std::array<uint32_t, 4> kw{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t k = 0; k < 4; ++k) {
      if (k >= nk) {  // an empty place: an inverted box and an entry the walk never follows
        r.entry[k] = 0xFFFFFFFFu;
        for (int a = 0; a < 3; ++a) q_lo[a][k] = 255u, q_hi[a][k] = 0u;
        continue;
      }
      const pt_packed_bvh_node& c = N[kids[k]];
      if (c.triangle_count == 0) {
        kw[k] = static_cast<uint32_t>(order.size());
        r.entry[k] = kw[k] << 7;
        order.push_back(kids[k]);
      } else {
        r.entry[k] = (c.left_or_begin << 7) | c.triangle_count;
      }
    }
    for (uint32_t k = 0; k < 4; ++k) {
      r.lo[0] |= q_lo[0][k] << (8 * k);
      r.lo[1] |= q_lo[1][k] << (8 * k);
      r.lo[2] |= q_lo[2][k] << (8 * k);
      r.hi_x |= q_hi[0][k] << (8 * k);
      r.hi_y |= q_hi[1][k] << (8 * k);
      r.hi_z |= q_hi[2][k] << (8 * k);
    }
    rec.push_back(r);
    kid_wide.push_back(kw);
    kid_count.push_back(nk);
  }
  // deferred children a walk may hold at once: entering one child of a record leaves up to (children - 1) on the stack
  std::vector<uint32_t> need(rec.size(), 0);
  for (size_t w = rec.size(); w-- > 0;) {
    uint32_t below = 0;
    for (uint32_t k = 0; k < 4; ++k)
      if (kid_wide[w][k] != 0xFFFFFFFFu && need[kid_wide[w][k]] > below) below = need[kid_wide[w][k]];
    need[w] = kid_count[w] - 1u + below;
  }
  stack_need = need.empty() ? 0u : need[0];
}

}  // namespace

extern "C" int pt_build_wide4(const pt_packed_bvh_node* nodes, uint32_t node_count, void* records_out, uint32_t capacity,
                              uint32_t* record_count_out, uint32_t* stack_need_out) {
  if (!nodes || !node_count || !record_count_out) return PT_ERR_INVALID;
  pt_scene_desc s{};
  s.bvh_nodes = nodes;
  s.bvh_node_count = node_count;
  std::vector<WideRecord> rec;
  uint32_t need = 0;
  build_wide(&s, rec, need);
  *record_count_out = static_cast<uint32_t>(rec.size());
  if (stack_need_out) *stack_need_out = need;
  if (records_out) {
    if (rec.size() > capacity) return PT_ERR_LIMIT;
    std::memcpy(records_out, rec.data(), rec.size() * sizeof(WideRecord));
  }
  return PT_OK;
}

static int upload_one(pt_ctx* ctx, const pt_scene_desc* s) {
  PT_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t depth = 0;
  int rc = validate_scene(ctx, s, &depth);
  if (rc != PT_OK) return rc;
  if (depth + 1 > 256)
    return fail(ctx, PT_ERR_LIMIT, "BVH depth %u exceeds the LDS traversal-stack limit of 255 levels", depth);

  PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_bufs(ctx->scene_bufs);
  ctx->has_scene = false;
  PT_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));

  pt_dev_scene d{};
  const size_t ntri = s->triangle_count;
  static_assert(sizeof(pt_packed_bvh_node) == 32 && sizeof(pt_material) == 64, "layout");
  // Child-pair records in breadth-first order: record p of inner node N is four quads
  // {lmin.x, rmin.x, lmin.y, rmin.y | lmin.z, rmin.z, lmax.x, rmax.x | lmax.y, rmax.y, lmax.z, rmax.z |
  //  left.ref, left.cnt, right.ref, right.cnt}, ref = the child's own record (inner) or its first
  // triangle (leaf). Boxes and triangle ranges are copied bit for bit from the reference array; only the
  // addressing and the order of the words change (left/right interleaved for the packed slab test).
  {
    const pt_packed_bvh_node* N = s->bvh_nodes;
    std::vector<uint32_t> order;  // inner nodes, breadth-first
    std::vector<uint32_t> record_of(s->bvh_node_count, 0xFFFFFFFFu);
    constexpr uint32_t kPad = 0xFFFFFFFFu;  // order[] entry of a padding record (sibling order only)
    const auto inner = [&](uint32_t c) { return N[c].triangle_count == 0; };
    if (N[0].triangle_count == 0) order.push_back(0);
    if (ctx->pair_order == 0 || order.empty()) {
      for (size_t head = 0; head < order.size(); ++head) {
        const uint32_t n = order[head];
        for (uint32_t side = 0; side < 2; ++side) {
          const uint32_t c = N[n].left_or_begin + side;
          if (inner(c)) order.push_back(c);
        }
      }
    } else {
      // Experiment orders (same records, other addresses: results identical by construction):
      //   1 depth-first, a record followed by everything under its left child (the reference's own numbering)
      //   2 the top `pair_order_levels` levels breadth-first, each sub-tree below them depth-first
      //   3 siblings adjacent (two inner children of a record next to each other, then everything under the left, then the right)
      //   4 as 3, with sibling pairs aligned to 128 bytes (a padding record where needed)
      order.clear();
      std::vector<uint32_t> roots{0u};
      if (ctx->pair_order == 2) {
        for (uint32_t level = 0; level < ctx->pair_order_levels; ++level) {
          std::vector<uint32_t> next;
          for (uint32_t n : roots) {
            order.push_back(n);
            for (uint32_t side = 0; side < 2; ++side)
              if (inner(N[n].left_or_begin + side)) next.push_back(N[n].left_or_begin + side);
          }
          roots.swap(next);
        }
      }
      std::vector<uint32_t> stack;
      for (uint32_t root : roots) {
        if (ctx->pair_order <= 2) {
          stack.assign(1, root);
          while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            order.push_back(n);
            const uint32_t l = N[n].left_or_begin;
            if (inner(l + 1)) stack.push_back(l + 1);
            if (inner(l)) stack.push_back(l);
          }
        } else {
          order.push_back(root);
          stack.assign(1, root);  // records already placed whose children are not
          while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            const uint32_t l = N[n].left_or_begin;
            if (ctx->pair_order == 4 && inner(l) && inner(l + 1) && (order.size() & 1u)) order.push_back(kPad);
            if (inner(l)) order.push_back(l);
            if (inner(l + 1)) order.push_back(l + 1);
            if (inner(l + 1)) stack.push_back(l + 1);
            if (inner(l)) stack.push_back(l);
          }
        }
      }
    }
    for (size_t p = 0; p < order.size(); ++p)
      if (order[p] != kPad) record_of[order[p]] = static_cast<uint32_t>(p);
    // record_of[] of a child is assigned when the child is dequeued, i.e. after its parent: two passes
    struct PairRecord {
      float box[12];
      uint32_t child[4];
    };
    static_assert(sizeof(PairRecord) == 64, "layout");
    std::vector<PairRecord> rec(order.size());
    for (size_t p = 0; p < order.size(); ++p) {
      if (order[p] == kPad) {
        std::memset(&rec[p], 0, sizeof(PairRecord));
        continue;
      }
      for (uint32_t side = 0; side < 2; ++side) {
        const uint32_t c = N[order[p]].left_or_begin + side;
        const pt_packed_bvh_node& r = N[c];
        for (int k = 0; k < 3; ++k) {
          rec[p].box[2 * k + side] = r.aabb_min[k];
          rec[p].box[6 + 2 * k + side] = r.aabb_max[k];
        }
        rec[p].child[2 * side] = r.triangle_count == 0 ? record_of[c] : r.left_or_begin;
        rec[p].child[2 * side + 1] = r.triangle_count;
      }
    }
    d.pair_count = static_cast<uint32_t>(order.size());
    // (ref << 7 | cnt) stack entries: every record and triangle index below 2^25, every leaf below 128 triangles
    d.packed_stack = (order.size() < (1u << 25) && ntri < (1u << 25)) ? 1u : 0u;
    for (size_t n = 0; n < s->bvh_node_count && d.packed_stack; ++n)
      if (N[n].triangle_count >= 128u) d.packed_stack = 0;
    d.root_ref = N[0].triangle_count ? N[0].left_or_begin : 0u;
    d.root_cnt = N[0].triangle_count;
    if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.pairs, reinterpret_cast<const float4*>(rec.data()), 4 * order.size())) != PT_OK)
      return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // rec dies here
  }
  // Triangle positions: the reference's binding 0 (3 x vec4 per triangle) with ONE word of its padding put to use: the w of a
  // triangle's first vertex is a tag, non-zero when the triangle's material is provably opaque (base_color_factor.a == 1 and no
  // base-colour texture or one whose image's alpha bytes are all 255: nearest returns the texel, linear lerps equal values). The
  // traversal has that quad in registers anyway, so a shadow walk knows without another fetch that its closest hit returns 0.
  std::vector<uint8_t> mat_opaque(s->material_count, 0);
  {
    std::vector<uint8_t> image_opaque(s->image_count, 0);  // 0 unknown, 1 all alpha bytes 255, 2 not
    for (uint32_t i = 0; i < s->material_count; ++i) {
      const pt_material& m = s->materials[i];
      if (m.base_color_factor[3] != 1.0f) continue;
      if (m.base_color_texture_index == -1) {
        mat_opaque[i] = 1;
        continue;
      }
      const int32_t ii = s->textures[m.base_color_texture_index].image_index;
      if (!image_opaque[ii]) {
        const pt_image& im = s->images[ii];
        const size_t n = static_cast<size_t>(im.width) * im.height;
        image_opaque[ii] = 1;
        for (size_t k = 0; k < n; ++k)
          if (im.rgba8[4 * k + 3] != 255) {
            image_opaque[ii] = 2;
            break;
          }
      }
      mat_opaque[i] = image_opaque[ii] == 1;
    }
  }
  {
    std::vector<float> pos(s->vertex_positions, s->vertex_positions + 12 * ntri);
    for (size_t t = 0; t < ntri; ++t) {
      const uint32_t tag = mat_opaque[static_cast<uint32_t>(s->packed_vertex_data[3 * t].material_index)] ? 1u : 0u;
      std::memcpy(&pos[12 * t + 3], &tag, 4);
    }
    if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.tri_pos, reinterpret_cast<const float4*>(pos.data()), 3 * ntri)) != PT_OK) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pos dies here
  }

  // 192-byte vertex triple -> 64-byte attribute record (drops tangent, bitangent, duplicate position: Q15)
  std::vector<float> attr(16 * ntri);
  for (size_t t = 0; t < ntri; ++t) {
    const pt_packed_vertex* v = s->packed_vertex_data + 3 * t;
    float* a = attr.data() + 16 * t;
    for (int k = 0; k < 3; ++k) std::memcpy(a + 3 * k, v[k].normal_and_texcoord_u, 12);
    for (int k = 0; k < 3; ++k) {
      a[9 + 2 * k] = v[k].normal_and_texcoord_u[3];
      a[10 + 2 * k] = v[k].tangent_and_texcoord_v[3];
    }
    std::memcpy(a + 15, &v[0].material_index, 4);
  }
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.tri_attr, reinterpret_cast<const float4*>(attr.data()), 4 * ntri)) != PT_OK)
    return rc;
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.materials, reinterpret_cast<const float4*>(s->materials),
                       4 * static_cast<size_t>(s->material_count))) != PT_OK)
    return rc;
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.directional_lights, reinterpret_cast<const float4*>(s->directional_lights),
                       2 * static_cast<size_t>(s->directional_light_count))) != PT_OK)
    return rc;
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.point_lights, reinterpret_cast<const float4*>(s->point_lights),
                       2 * static_cast<size_t>(s->point_light_count))) != PT_OK)
    return rc;
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.emissive_triangles, s->emissive_triangles,
                       static_cast<size_t>(s->emissive_triangle_count))) != PT_OK)
    return rc;

  // textures[] -> one record per combined image sampler; images -> one texel pool
  std::vector<uint64_t> image_offset(s->image_count);
  uint64_t total_texels = 0;
  for (uint32_t i = 0; i < s->image_count; ++i) {
    image_offset[i] = total_texels;
    total_texels += static_cast<uint64_t>(s->images[i].width) * s->images[i].height;
  }
  if (total_texels > 0xFFFFFFFFull) return fail(ctx, PT_ERR_LIMIT, "texture pool exceeds 2^32 texels");
  std::vector<uint32_t> pool(total_texels);
  for (uint32_t i = 0; i < s->image_count; ++i)
    std::memcpy(pool.data() + image_offset[i], s->images[i].rgba8, 4 * static_cast<size_t>(s->images[i].width) * s->images[i].height);
  std::vector<uint4> trec(s->texture_count);
  for (uint32_t i = 0; i < s->texture_count; ++i) {
    const pt_image& im = s->images[s->textures[i].image_index];
    const pt_sampler& sm = s->samplers[s->textures[i].sampler_index];
    auto wrap = [](int32_t w) { return (w == PT_WRAP_CLAMP_TO_EDGE || w == PT_WRAP_MIRRORED_REPEAT) ? static_cast<uint32_t>(w) : 2u; };
    uint32_t flags = (sm.mag_filter == PT_FILTER_NEAREST ? 0u : 1u) | (wrap(sm.wrap_s) << 2) | (wrap(sm.wrap_t) << 4);
    trec[i] = make_uint4(static_cast<uint32_t>(image_offset[s->textures[i].image_index]), im.width, im.height, flags);
  }
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.texels, pool.data(), pool.size())) != PT_OK) return rc;
  if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.textures, trec.data(), trec.size())) != PT_OK) return rc;

  // Provably all-opaque scene: every material is (shadow rays may then stop at the first accepted triangle)
  d.all_opaque = 1;
  for (uint32_t i = 0; i < s->material_count; ++i)
    if (!mat_opaque[i]) d.all_opaque = 0;
  d.any_texture = 0;
  for (uint32_t i = 0; i < s->material_count; ++i) {
    const pt_material& m = s->materials[i];
    if (m.base_color_texture_index != -1 || m.metallic_roughness_texture_index != -1 || m.emissive_texture_index != -1 ||
        m.occlusion_texture_index != -1)
      d.any_texture = 1;
  }
  d.directional_light_count = s->directional_light_count;
  d.point_light_count = s->point_light_count;
  d.stack_entries = depth + 1;
  d.triangle_count = s->triangle_count;
  d.material_count = s->material_count;
  d.emissive_count = s->emissive_triangle_count;
  // LDS residency: a scene whose pair records + triangle positions fit in lds_scene_bytes
  // is staged whole in every traversal workgroup's LDS (next to the 1 KiB-per-level stack).
  d.scene_in_lds = (64ull * d.pair_count + (48ull + 64ull) * s->triangle_count + 64ull * s->material_count +
                        32ull * (s->directional_light_count + s->point_light_count) + 4ull * s->emissive_triangle_count <=
                    ctx->lds_scene_bytes &&
                    d.pair_count < 65536u && s->triangle_count < 65536u)  // LDS stack entries pack (ref << 16 | cnt)
                       ? 1u : 0u;
  if (ctx->wide4 && !d.scene_in_lds && d.packed_stack) {  // PT_BVH=wide4 (opt-in): the collapsed tree beside the reference's
    std::vector<WideRecord> wide;
    build_wide(s, wide, d.wide_stack_need);
    d.wide_count = static_cast<uint32_t>(wide.size());
    if (d.wide_count && d.wide_count < (1u << 25)) {
      if ((rc = dev_upload(ctx, ctx->scene_bufs, &d.wide, reinterpret_cast<const float4*>(wide.data()), 4 * wide.size())) != PT_OK) return rc;
      PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // wide dies here
    }
  }
  PT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors die here
  float ms = 0.0f;
  PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
  ctx->upload_ms = ms;
  ctx->scene = d;
  ctx->emissive_count = s->emissive_triangle_count;
  ctx->has_scene = true;
  return PT_OK;
}

namespace {

hipEvent_t next_kernel_event(pt_ctx* ctx, size_t& used) {
  if (used == ctx->kernel_events.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    ctx->kernel_events.push_back(e);
  }
  return ctx->kernel_events[used++];
}

// no_alloc: the frame's working set is the file-scope scratch array (ctx->scratch_state) and nothing is allocated on the way:
// path pipeline only (the wavefront pipeline needs request buffers sized by the light count), no per-kernel timing
// (its events are created on demand), outputs straight into the caller's buffers.
int render_accum_impl(pt_ctx* ctx, const pt_render_params* p, float4* accum_dev, uint32_t* primary_dev, hipStream_t stream,
                      pt_stats* stats_out, bool no_alloc) {
  if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "pt_upload_scene has not been called");
  const pt_push_constants& pc = p->pc;
  if (!pc.resolution_x || !pc.resolution_y || pc.resolution_x > 65536 || pc.resolution_y > 65536)
    return fail(ctx, PT_ERR_INVALID, "resolution %ux%u out of range", pc.resolution_x, pc.resolution_y);
  if (pc.emissive_triangle_count != ctx->emissive_count)
    return fail(ctx, PT_ERR_INVALID, "push constants say %u emissive triangles, the uploaded scene has %u",
                pc.emissive_triangle_count, ctx->emissive_count);
  const uint32_t num_ranks = p->num_ranks ? p->num_ranks : 1;
  if (p->rank >= num_ranks) return fail(ctx, PT_ERR_INVALID, "rank %u >= num_ranks %u", p->rank, num_ranks);

  pt_dev_frame f{};
  f.pc = pc;
  for (int k = 0; k < 3; ++k) f.bg_linear[k] = pt_pow(pc.bg_color[k], 2.2f);
  f.sqrt_samples = static_cast<uint32_t>(static_cast<int32_t>(std::sqrt(static_cast<float>(pc.sample_count))));
  const uint32_t total = f.sqrt_samples * f.sqrt_samples;
  if (f.sqrt_samples == 0 || total >= (1u << 24))
    return fail(ctx, PT_ERR_LIMIT, "sample_count %u gives %u strata; supported range is 1 .. 2^24-1", pc.sample_count, total);
  f.rank = p->rank;
  f.num_ranks = num_ranks;
  f.sample_begin = static_cast<uint32_t>((static_cast<uint64_t>(p->rank) * total) / num_ranks);
  f.sample_end = static_cast<uint32_t>((static_cast<uint64_t>(p->rank + 1) * total) / num_ranks);
  f.tiles_x = (pc.resolution_x + 7) / 8;
  const uint32_t tiles_y = (pc.resolution_y + 7) / 8;
  const uint64_t slots64 = static_cast<uint64_t>(f.tiles_x) * tiles_y * 64;
  if (slots64 > 0x7FFFFFFFull) return fail(ctx, PT_ERR_LIMIT, "image too large");
  const uint32_t total_slots = static_cast<uint32_t>(slots64);
  const uint32_t pixels = pc.resolution_x * pc.resolution_y;

  // The frame is rendered in BANDS of pixel slots so that the per-slot working set (rays,
  // hits, path state, shadow requests, queues: ~220 B + 64 B per light per slot) stays
  // resident in the 256 MiB Infinity Cache instead of streaming through HBM every
  // iteration; a band is still several waves per SIMD of work (DESIGN.md "Banding").
  pt_dev_scene sc = ctx->scene;
  if (!ctx->child_entries) sc.packed_stack = 0;
  const uint32_t n_lights = sc.directional_light_count + sc.point_light_count + (pc.emissive_triangle_count ? 1u : 0u);
  uint32_t band_slots = ctx->band_slots ? ctx->band_slots : total_slots;
  band_slots = (band_slots + 63u) & ~63u;
  if (band_slots > total_slots) band_slots = total_slots;
  // traversal block size: stack (block * stack_entries words) + the scene copy, kept within 64 KiB
  const uint32_t scene_lds = sc.scene_in_lds ? 64u * sc.pair_count + (48u + 64u) * sc.triangle_count + 64u * sc.material_count +
                                                   32u * (sc.directional_light_count + sc.point_light_count) + 4u * sc.emissive_count + 16u
                                             : 0u;
  if (256u * sc.stack_entries * 4u + scene_lds > 65536u) sc.scene_in_lds = 0;  // deep tree: the stack needs the LDS
  uint32_t trace_block = (256u * sc.stack_entries * 4u <= 65536u) ? 256u : 64u;
  const int pipeline = no_alloc ? 0 : ctx->pipeline;
  // k_paths_merged: the path pipeline on a scene read from global memory whose vertices draw at most two light samples (its two stash slots)
  const bool merged = pipeline == 0 && ctx->merged && !sc.scene_in_lds && n_lights <= 2u;
  // (one light sample per vertex: 40 bytes of stash per lane instead of 80, so 16 rows fit where 9 do with two — rocks 64 spp, rows
  // 9 / 11 / 13 / 16 / 19 / 20: 1672 / 1700 / 1709 / 1710 / 1707 / 1609 Mrays/s)
  const uint32_t lds_stack = merged ? (ctx->merged_lds_stack ? ctx->merged_lds_stack - 1u : (ptd::merged_stash_samples(sc, n_lights) > 1u ? 9u : 16u)) : ctx->lds_stack;
  static_assert(PT_DEEP_LEVELS == 48, "the private overflow rows of traverse() (pt_kernels.hip, kDeepLevels) bound the inequality below");
  if (!sc.scene_in_lds && lds_stack && lds_stack < sc.stack_entries && sc.stack_entries <= lds_stack + PT_DEEP_LEVELS) {
    sc.lds_stack_levels = lds_stack;  // short LDS stack, the rest of the depth in private memory (kDeepLevels = 48 rows)
    trace_block = 256u;
  }
  // The traversal kernels exist in two compilations (pt_kernels.hip, PT_STACK_CHILD): stack entries that hold the deferred child
  // itself for scenes that allow it, (record, side) entries otherwise. LDS-resident scenes use neither format (their own).
  const bool rs = !sc.scene_in_lds && !sc.packed_stack;
  const auto launch_extend = rs ? ptd_rs::launch_extend : ptd::launch_extend;
  const auto launch_shade = rs ? ptd_rs::launch_shade : ptd::launch_shade;
  const auto launch_shadow = rs ? ptd_rs::launch_shadow : ptd::launch_shadow;
  // PT_BVH=wide4: only k_paths_merged of the exact compilation walks the wide records, and only if the lane's LDS rows plus its 128
  // private rows hold the worst-case number of deferred children; everything else walks the reference's tree.
  if (sc.wide && !(merged && !rs &&
                   sc.wide_stack_need + 1u <= (sc.lds_stack_levels ? sc.lds_stack_levels : sc.stack_entries) + 128u))
    sc.wide = nullptr;
  const auto launch_paths_exact = rs ? ptd_rs::launch_paths : ptd::launch_paths;
  // PT_ARITH=fast applies to scenes staged in LDS, which are bound by VALU issue (cfg2 +13 %). Scenes read from global memory wait on
  // memory: there the fast compilation never gained (round 2: 71.8 against 69.4 ms per launch on the atrium) and round 3 measured it
  // SLOWER (its register allocation spills inside the walk loops: atrium 1256 / 1714 Mrays/s merged / k_paths against 2052 / 1878), so
  // those scenes run the exact kernels whatever PT_ARITH says — and stay bit-exact.
  const bool fast = ctx->fast_arith && sc.scene_in_lds;
  const auto launch_paths = merged ? (rs ? ptd_rs::launch_paths_merged : ptd::launch_paths_merged) : (fast ? ptd_fast::launch_paths : launch_paths_exact);
  // Concurrent bands (path pipeline): the frame is cut into `conc` bands of tiles, each with its own state, queues and stream,
  // and their launches are enqueued alternately. Every k_paths launch ends with a tail in which most of the GPU idles while the
  // last waves finish their vertices (a 1080p launch is only ~5 rounds of resident waves); the other band's launch fills it.
  // Per slot nothing changes, so the frame is the same to the bit.
  // Per-kernel timing (PT_FLAG_TIME_KERNELS) renders the bands one after another: the duration of a launch that shares the GPU
  // with another band's launch says nothing about the kernel.
  const bool time_kernels = (p->flags & PT_FLAG_TIME_KERNELS) != 0;
  uint32_t conc = 1;
  if (pipeline == 0 && !no_alloc && !time_kernels && ctx->concurrent_bands > 1 && band_slots == total_slots) {
    const uint32_t tiles = total_slots / 64;
    conc = ctx->concurrent_bands;
    while (conc > 1 && tiles / conc < ctx->band_min_tiles) --conc;  // a band should still fill the GPU (4096 waves resident) on its own
    if (conc > 1) band_slots = ((tiles + conc - 1) / conc) * 64u;
  }
  const bool split = pipeline == 1 && ctx->split_shadow && !sc.scene_in_lds && n_lights != 0;
  int rc = PT_OK;
  if (no_alloc) {
    if (!ctx->scratch) return fail(ctx, PT_ERR_DEVICE, "no scratch array on device %d", ctx->device);
    if (total_slots > kNoAllocSlots)
      return fail(ctx, PT_ERR_LIMIT,
                  "pt_render_accum_device_noalloc: %ux%u needs %u pixel slots (8x8 tiles), the fixed scratch holds %u "
                  "(e.g. 512x512); use pt_render_accum_device",
                  pc.resolution_x, pc.resolution_y, total_slots, kNoAllocSlots);
    if (p->flags & PT_FLAG_TIME_KERNELS)
      return fail(ctx, PT_ERR_INVALID, "pt_render_accum_device_noalloc: PT_FLAG_TIME_KERNELS creates events on demand; not available here");
    band_slots = total_slots;
  } else {
    rc = ensure_state(ctx, band_slots, split ? n_lights : 0u, pixels);
    if (rc != PT_OK) return rc;
    if (conc > 1 && (rc = ensure_bands(ctx, conc, band_slots)) != PT_OK) return rc;
    if (!accum_dev) accum_dev = ctx->accum_out;
    if (!primary_dev) primary_dev = ctx->primary_out;
  }
  const pt_dev_state& st = no_alloc ? ctx->scratch_state : ctx->state;
  size_t ev_used = 0;
  std::vector<std::pair<size_t, int>> ev_kind;  // index of the start event, kernel id
  uint64_t launches[PT_K_COUNT + 1] = {0, 0, 0, 0, 0};

  auto timed_on = [&](hipStream_t on, int kind, auto&& launch) {
    ++launches[kind];
    if (ctx->log_launches) std::fprintf(stderr, "[pt] launch of kernel kind %d (1 extend, 2 shade, 3 shadow, 4 paths)\n", kind);
    if (time_kernels) {
      hipEvent_t a = next_kernel_event(ctx, ev_used);
      hipEvent_t b = next_kernel_event(ctx, ev_used);
      if (a && b) {
        ev_kind.emplace_back(ev_used - 2, kind);
        (void)hipEventRecord(a, on);
        launch();
        (void)hipEventRecord(b, on);
        return;
      }
    }
    launch();
  };
  auto timed = [&](int kind, auto&& launch) { timed_on(stream, kind, launch); };

  PT_HIP(ctx, hipMemsetAsync(st.stats, 0, static_cast<size_t>(PT_K_COUNT) * st.stat_rows * PT_ST_WORDS * sizeof(unsigned long long), stream));
  PT_HIP(ctx, hipEventRecord(ctx->ev_start, stream));

  // One path needs at most 8 light steps + 4 pass-through vertices (main.comp:18-19, 512, 542).
  const uint64_t max_iters = static_cast<uint64_t>(f.sample_end - f.sample_begin) * 12ull + 2ull;
  constexpr uint32_t kBatch = 16;
  uint64_t iter_total = 0;
  if (conc > 1) {
    struct Run {
      const pt_dev_state* st;
      hipStream_t s;
      hipEvent_t* pe;
      hipEvent_t done;
      uint32_t* hc;
      pt_dev_frame f;
      uint32_t bound, k;
      uint64_t vertices;
      int poll;
      bool have_prev, drained;
    };
    std::vector<Run> runs;
    // An error return from inside this block must not leave band streams running: their launches write the caller's buffers, which the
    // caller may free once the call has failed. (The success path joins through events instead; the host does not wait.)
    struct Quiesce {
      std::vector<Run>& runs;
      bool joined = false;
      ~Quiesce() {
        if (!joined)
          for (Run& r : runs) (void)hipStreamSynchronize(r.s);
      }
    } quiesce{runs};
    PT_HIP(ctx, hipEventRecord(ctx->fork_event, stream));
    for (uint32_t c = 0; c < conc && c * band_slots < total_slots; ++c) {
      pt_ctx::Band* band = &ctx->bands[c];
      Run r{};
      r.st = c ? &band->state : &ctx->state;
      r.s = band->stream;
      r.pe = band->poll_events;
      r.done = band->done;
      r.hc = band->host_counts;
      r.f = f;
      r.f.slot_base = c * band_slots;
      r.f.slot_count = (total_slots - r.f.slot_base < band_slots) ? total_slots - r.f.slot_base : band_slots;
      r.bound = r.f.slot_count;
      r.drained = (f.sample_begin == f.sample_end);
      PT_HIP(ctx, hipStreamWaitEvent(r.s, ctx->fork_event, 0));
      if (c) PT_HIP(ctx, hipMemsetAsync(r.st->stats, 0, static_cast<size_t>(PT_K_COUNT) * r.st->stat_rows * PT_ST_WORDS * sizeof(unsigned long long), r.s));
      PT_HIP(ctx, hipMemsetAsync(r.st->counters, 0, PT_CNT_WORDS * sizeof(uint32_t), r.s));
      ptd::launch_init(r.f, *r.st, r.s);
      runs.push_back(r);
    }
    for (bool any = true; any;) {
      any = false;
      for (Run& r : runs) {
        if (r.drained || r.vertices >= max_iters) continue;
        any = true;
        timed_on(r.s, PT_K_PATHS, [&] {
          launch_paths(
              sc, r.f, *r.st, r.k & 1u, PT_CNT_QUEUE0 + r.k % 3u, PT_CNT_QUEUE0 + (r.k + 1u) % 3u, PT_CNT_QUEUE0 + (r.k + 2u) % 3u, r.bound,
              trace_block < ctx->paths_block_threads ? trace_block : ctx->paths_block_threads, ctx->paths_blocks, ctx->paths_vertices,
              sc.scene_in_lds ? 0u : ctx->xcd_chunk, r.s);
        });
        PT_HIP(ctx, hipGetLastError());
        // the band's live count after this launch, looked at one launch later (the GPU never waits for the host)
        PT_HIP(ctx, hipMemcpyAsync(&r.hc[r.poll], r.st->counters + PT_CNT_QUEUE0 + (r.k + 1u) % 3u, sizeof(uint32_t), hipMemcpyDeviceToHost, r.s));
        PT_HIP(ctx, hipEventRecord(r.pe[r.poll], r.s));
        if (r.have_prev) {
          PT_HIP(ctx, hipEventSynchronize(r.pe[r.poll ^ 1]));
          const uint32_t live = r.hc[r.poll ^ 1];
          if (ctx->log_launches) std::fprintf(stderr, "[pt] band %u: %u slots live after the launch before last\n", r.f.slot_base, live);
          if (live == 0) r.drained = true;
          else if (live < r.bound) r.bound = live;
        }
        r.have_prev = true;
        r.poll ^= 1;
        ++r.k;
        r.vertices += ctx->paths_vertices;
      }
    }
    for (Run& r : runs) {
      if (!r.drained) {
        PT_HIP(ctx, hipEventSynchronize(r.pe[r.poll ^ 1]));
        if (r.hc[r.poll ^ 1] != 0)
          return fail(ctx, PT_ERR_DEVICE, "render loop did not drain after %u launches", r.k);
      }
      ptd::launch_finalize(r.f, *r.st, accum_dev, primary_dev, r.s);
      if (stats_out) ptd::launch_sum_stats(*r.st, r.s);
      PT_HIP(ctx, hipEventRecord(r.done, r.s));  // join: the caller's stream goes on after every band
      PT_HIP(ctx, hipStreamWaitEvent(stream, r.done, 0));
      if (r.k > iter_total) iter_total = r.k;
    }
    quiesce.joined = true;
  } else
  for (uint32_t base = 0; base < total_slots; base += band_slots) {
    f.slot_base = base;
    f.slot_count = (total_slots - base < band_slots) ? total_slots - base : band_slots;
    PT_HIP(ctx, hipMemsetAsync(st.counters, 0, PT_CNT_WORDS * sizeof(uint32_t), stream));
    ptd::launch_init(f, st, stream);

    uint32_t bound = f.slot_count;  // live entries never increase: the last polled count bounds every later queue
    uint64_t iter = 0;
    int poll = 0;
    bool have_prev = false;
    bool drained = (f.sample_begin == f.sample_end);
    const uint32_t max_vertices_u32 = static_cast<uint32_t>(max_iters > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_iters);
    // look at the previous poll while the launches just enqueued run: the GPU never waits for the host
    auto poll_counts = [&](uint32_t counter_word) -> int {
      PT_HIP(ctx, hipMemcpyAsync(&ctx->host_counts[poll], st.counters + counter_word, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      PT_HIP(ctx, hipEventRecord(ctx->poll_events[poll], stream));
      if (have_prev) {
        PT_HIP(ctx, hipEventSynchronize(ctx->poll_events[poll ^ 1]));
        const uint32_t live = ctx->host_counts[poll ^ 1];
        if (ctx->log_launches) std::fprintf(stderr, "[pt] band %u: %u slots live after the launch before last\n", base, live);
        if (live == 0) drained = true;
        else if (live < bound) bound = live;
      }
      have_prev = true;
      poll ^= 1;
      return PT_OK;
    };

    if (pipeline == 0) {
      // ---- path pipeline: one k_paths launch advances every queued slot by up to paths_vertices vertices
      uint64_t vertices = 0;
      for (uint32_t k = 0; !drained && vertices < max_iters; ++k, vertices += ctx->paths_vertices) {
        timed(PT_K_PATHS, [&] {
          launch_paths(
                            sc, f, st, k & 1u, PT_CNT_QUEUE0 + k % 3u, PT_CNT_QUEUE0 + (k + 1u) % 3u, PT_CNT_QUEUE0 + (k + 2u) % 3u, bound,
                            trace_block < ctx->paths_block_threads ? trace_block : ctx->paths_block_threads, ctx->paths_blocks,
                            ctx->paths_vertices, sc.scene_in_lds ? 0u : ctx->xcd_chunk, stream);
        });
        PT_HIP(ctx, hipGetLastError());
        if ((rc = poll_counts(PT_CNT_QUEUE0 + (k + 1u) % 3u)) != PT_OK) return rc;
        ++iter;
      }
    } else {
      // ---- wavefront pipeline: k_extend + k_shade (+ k_shadow) per path vertex
      uint32_t parity = 0;  // the queue the next iteration reads
      while (!drained && iter < max_iters) {
        for (uint32_t b = 0; b < kBatch; ++b, ++iter) {
          // the queue is rebuilt on the last iteration of a batch (every iteration in the split pipeline,
          // whose shadow queue is an append anyway) and re-used as it stands in between
          const bool compact = split || ctx->compact_every <= 1 || (b + 1) % ctx->compact_every == 0 || b + 1 == kBatch;
          timed(PT_K_EXTEND, [&] { launch_extend(sc, st, parity, bound, trace_block, ctx->trace_blocks, stream); });
          timed(PT_K_SHADE, [&] {
            launch_shade(sc, f, st, parity, bound, trace_block, ctx->shade_blocks, n_lights != 0, split, compact, stream);
          });
          if (split) timed(PT_K_SHADOW, [&] { launch_shadow(sc, f, st, bound, trace_block, ctx->trace_blocks, stream); });
          if (compact) parity ^= 1u;
        }
        PT_HIP(ctx, hipGetLastError());
        // entries queued for the NEXT iteration; stays intact until that iteration's successor recycles it
        if ((rc = poll_counts(PT_CNT_QUEUE0 + parity)) != PT_OK) return rc;
        if (!drained && bound <= ctx->tail_entries) {
          // short queue: every remaining vertex of every queued slot in one k_paths launch
          timed(PT_K_PATHS, [&] {
            launch_paths_exact(sc, f, st, parity, PT_CNT_QUEUE0 + parity, PT_CNT_QUEUE2, PT_CNT_QUEUE0 + (parity ^ 1u), bound, trace_block,
                              0xFFFFFFFFu, max_vertices_u32, 0u, stream);
          });
          PT_HIP(ctx, hipGetLastError());
          drained = true;
        }
      }
    }
    if (!drained) {
      PT_HIP(ctx, hipEventSynchronize(ctx->poll_events[poll ^ 1]));
      if (ctx->host_counts[poll ^ 1] != 0)
        return fail(ctx, PT_ERR_DEVICE, "render loop did not drain after %llu launches", static_cast<unsigned long long>(iter));
    }
    ptd::launch_finalize(f, st, accum_dev, primary_dev, stream);
    iter_total += iter;
  }
  const uint64_t iter = iter_total;
  if (stats_out && conc == 1) ptd::launch_sum_stats(st, stream);
  PT_HIP(ctx, hipEventRecord(ctx->ev_stop, stream));
  PT_HIP(ctx, hipGetLastError());
  PT_HIP(ctx, hipStreamSynchronize(stream));

  if (stats_out) {
    unsigned long long h[PT_K_COUNT][PT_ST_WORDS] = {};
    // k_sum_stats ran at the end of the frame; read its 256 bytes through the context's pinned words (no staging allocation)
    unsigned long long* pinned = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(ctx->host_counts) + 64);
    PT_HIP(ctx, hipMemcpyAsync(pinned, st.stat_totals, sizeof(h), hipMemcpyDeviceToHost, stream));
    PT_HIP(ctx, hipStreamSynchronize(stream));
    std::memcpy(&h[0][0], pinned, sizeof(h));
    for (uint32_t c = 1; c < conc && c * band_slots < total_slots; ++c) {  // the other concurrent bands' totals
      pt_ctx::Band& band = ctx->bands[c];
      unsigned long long* bp = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(band.host_counts) + 64);
      PT_HIP(ctx, hipMemcpyAsync(bp, band.state.stat_totals, sizeof(h), hipMemcpyDeviceToHost, stream));
      PT_HIP(ctx, hipStreamSynchronize(stream));
      for (int k = 0; k < PT_K_COUNT; ++k)
        for (int w = 0; w < PT_ST_WORDS; ++w) h[k][w] += bp[k * PT_ST_WORDS + w];
    }
    std::memset(stats_out, 0, sizeof(*stats_out));
    auto total = [&](int w) { return h[PT_K_INIT][w] + h[PT_K_EXTEND][w] + h[PT_K_SHADE][w] + h[PT_K_SHADOW][w]; };
    // (the shadow slice's PATHS word counts the shortcut-decided shadow rays, not paths)
    stats_out->paths = h[PT_K_INIT][PT_ST_PATHS] + h[PT_K_EXTEND][PT_ST_PATHS] + h[PT_K_SHADE][PT_ST_PATHS];
    stats_out->rays_closest = h[PT_K_EXTEND][PT_ST_RAYS];
    stats_out->rays_shadow_shortcut = h[PT_K_SHADOW][PT_ST_PATHS];
    stats_out->rays_shadow_walked = h[PT_K_SHADE][PT_ST_RAYS] + h[PT_K_SHADOW][PT_ST_RAYS] - h[PT_K_SHADOW][PT_ST_PATHS];
    stats_out->rays = total(PT_ST_RAYS);
    stats_out->node_visits = total(PT_ST_VISITS);
    stats_out->tri_tests = total(PT_ST_TESTS);
    stats_out->hits_shaded = total(PT_ST_HITS);
    stats_out->texture_taps = total(PT_ST_TAPS);
    stats_out->shaded_vertices = total(PT_ST_VERTS);
    stats_out->nan_samples = total(PT_ST_NANS);
    stats_out->iterations = iter;
    stats_out->scene_in_lds = sc.scene_in_lds;
    stats_out->merged_walk = merged ? 1u : 0u;
    stats_out->bvh_wide4 = sc.wide ? 1u : 0u;
    stats_out->fast_arith = fast ? 1u : 0u;
    pt_kernel_stats* per[PT_K_COUNT + 1] = {nullptr, &stats_out->extend, &stats_out->shade, &stats_out->shadow, &stats_out->path_kernel};
    stats_out->path_kernel.launches = launches[PT_K_PATHS];
    for (int k = PT_K_EXTEND; k < PT_K_COUNT; ++k) {
      per[k]->launches = launches[k];
      per[k]->rays = h[k][PT_ST_RAYS];
      per[k]->node_visits = h[k][PT_ST_VISITS];
      per[k]->tri_tests = h[k][PT_ST_TESTS];
      per[k]->hits_shaded = h[k][PT_ST_HITS];
      per[k]->texture_taps = h[k][PT_ST_TAPS];
    }
    float ms = 0.0f;
    PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    stats_out->render_ms = ms;
    stats_out->upload_ms = ctx->upload_ms;
    for (auto& [idx, kind] : ev_kind) {
      float k = 0.0f;
      if (hipEventElapsedTime(&k, ctx->kernel_events[idx], ctx->kernel_events[idx + 1]) == hipSuccess) per[kind]->ms += k;
      if (ctx->log_launches) std::fprintf(stderr, "[pt] launch %zu: kernel %d, %.3f ms\n", idx / 2, kind, k);
    }
  }
  return PT_OK;
}

// The call is blocking when it succeeds (it ends with a stream synchronisation); a call that fails half-way must not return while
// launches it enqueued still write the caller's buffers.
int render_accum(pt_ctx* ctx, const pt_render_params* p, float4* accum_dev, uint32_t* primary_dev, hipStream_t stream,
                 pt_stats* stats_out, bool no_alloc = false) {
  const int rc = render_accum_impl(ctx, p, accum_dev, primary_dev, stream, stats_out, no_alloc);
  if (rc != PT_OK) (void)hipStreamSynchronize(stream);
  return rc;
}

}  // namespace

extern "C" int pt_render_accum_device(pt_ctx* ctx, const pt_render_params* params, float* accum_dev,
                                      uint32_t* primary_id_dev, void* stream, pt_stats* stats_out) {
  if (!ctx) return PT_ERR_INVALID;
  if (!params || !accum_dev) return fail(ctx, PT_ERR_INVALID, "params / accum_dev is NULL");
  PT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  return render_accum(ctx, params, reinterpret_cast<float4*>(accum_dev), primary_id_dev, s, stats_out);
}

extern "C" int pt_render_accum_device_noalloc(pt_ctx* ctx, const pt_render_params* params, float* accum_dev,
                                              uint32_t* primary_id_dev, void* stream, pt_stats* stats_out) {
  if (!ctx) return PT_ERR_INVALID;
  if (!params || !accum_dev) return fail(ctx, PT_ERR_INVALID, "params / accum_dev is NULL");
  PT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  return render_accum(ctx, params, reinterpret_cast<float4*>(accum_dev), primary_id_dev, s, stats_out, /*no_alloc=*/true);
}

extern "C" int pt_resolve_device(pt_ctx* ctx, const float* accum_dev, uint32_t pixel_count, uint8_t* rgba8_dev,
                                 void* stream) {
  if (!ctx) return PT_ERR_INVALID;
  if (!accum_dev || !rgba8_dev || !pixel_count) return fail(ctx, PT_ERR_INVALID, "resolve: NULL buffer or no pixels");
  PT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  ptd::launch_resolve(reinterpret_cast<const float4*>(accum_dev), pixel_count, reinterpret_cast<uint32_t*>(rgba8_dev), s);
  PT_HIP(ctx, hipGetLastError());
  return PT_OK;
}

extern "C" pt_ctx* pt_create_multi(const int* device_ids, int n) {
  if (!device_ids || n < 1 || n > kMaxDevices) {
    fail(nullptr, PT_ERR_INVALID, "pt_create_multi: need 1..%d device ids", kMaxDevices);
    return nullptr;
  }
  std::vector<int> ids(device_ids, device_ids + n);
  for (int& id : ids)
    if (id < 0) id = 0;  // pt_create's rule: a negative id picks GPU 0 — applied BEFORE the duplicate check, so {-1, 0} is refused
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (ids[i] == ids[j]) {
        fail(nullptr, PT_ERR_INVALID, "pt_create_multi: device %d listed twice", ids[i]);
        return nullptr;
      }
  RcclApi& api = rccl();
  if (!api.error.empty()) {
    fail(nullptr, PT_ERR_DEVICE, "pt_create_multi: %s", api.error.c_str());
    return nullptr;
  }
  std::vector<pt_ctx*> members;
  for (int i = 0; i < n; ++i) {
    pt_ctx* m = pt_create(ids[i]);
    if (!m) {
      for (pt_ctx* x : members) pt_destroy(x);
      return nullptr;  // pt_create left its message
    }
    members.push_back(m);
  }
  std::vector<int> devs;
  for (pt_ctx* m : members) devs.push_back(m->device);
  std::vector<ncclComm_t> comms(static_cast<size_t>(n), nullptr);
  const ncclResult_t r = api.CommInitAll(comms.data(), n, devs.data());  // one communicator per GPU, this process owns all ranks
  if (r != ncclSuccess) {
    fail(nullptr, PT_ERR_DEVICE, "ncclCommInitAll over %d device(s) failed: %s", n, api.GetErrorString(r));
    for (pt_ctx* x : members) pt_destroy(x);
    return nullptr;
  }
  pt_ctx* root = members[0];
  root->members = members;
  root->comms = comms;
  return root;
}

namespace {

// pt_render over a device group: rank i renders strata [i n^2/R, (i+1) n^2/R) of every pixel on its own GPU (one host
// thread per GPU: the render loop polls its stream), then ONE ncclReduce(sum, float32) of the float4 accumulators to
// rank 0 over xGMI, which resolves. R = 1 runs the same code (a one-rank communicator), which is how a single-GPU box tests it.
int render_group(pt_ctx* root, const pt_render_params* params, pt_stats* stats_out) {
  const size_t R = root->members.size();
  if (params->num_ranks > 1)
    return fail(root, PT_ERR_INVALID, "a device group splits the strata itself: pass rank 0 / num_ranks 0 or 1, not %u / %u",
                params->rank, params->num_ranks);
  std::vector<pt_render_params> ps(R, *params);
  std::vector<pt_stats> sts(R);
  std::vector<int> rcs(R, PT_OK);
  std::vector<std::thread> threads;
  for (size_t i = 0; i < R; ++i) {
    ps[i].rank = static_cast<uint32_t>(i);
    ps[i].num_ranks = static_cast<uint32_t>(R);
    threads.emplace_back([&, i] {
      pt_ctx* m = root->members[i];
      if (hipSetDevice(m->device) != hipSuccess) {
        rcs[i] = fail(m, PT_ERR_DEVICE, "hipSetDevice(%d) failed", m->device);
        return;
      }
      rcs[i] = render_accum(m, &ps[i], nullptr, nullptr, m->stream, &sts[i]);
    });
  }
  for (std::thread& t : threads) t.join();
  for (size_t i = 0; i < R; ++i)
    if (rcs[i] != PT_OK) {
      if (i) root->err = "device " + std::to_string(root->members[i]->device) + ": " + root->members[i]->err;
      return rcs[i];
    }
  // the path's one exchange step
  RcclApi& api = rccl();
  const size_t count = 4 * static_cast<size_t>(params->pc.resolution_x) * params->pc.resolution_y;
  PT_HIP(root, hipSetDevice(root->device));
  PT_HIP(root, hipEventRecord(root->ev_start, root->stream));
  ncclResult_t r = api.GroupStart();
  for (size_t i = 0; i < R && r == ncclSuccess; ++i) {
    pt_ctx* m = root->members[i];
    r = api.Reduce(m->accum_out, m->accum_out, count, ncclFloat32, ncclSum, 0, root->comms[i], m->stream);  // in place; only rank 0's buffer receives
  }
  const ncclResult_t e = api.GroupEnd();
  if (r == ncclSuccess) r = e;
  if (r != ncclSuccess) return fail(root, PT_ERR_DEVICE, "ncclReduce of the accumulators failed: %s", api.GetErrorString(r));
  PT_HIP(root, hipSetDevice(root->device));
  PT_HIP(root, hipEventRecord(root->ev_stop, root->stream));
  for (pt_ctx* m : root->members) {
    PT_HIP(root, hipSetDevice(m->device));
    PT_HIP(root, hipStreamSynchronize(m->stream));
  }
  PT_HIP(root, hipSetDevice(root->device));
  float ms = 0.0f;
  PT_HIP(root, hipEventElapsedTime(&ms, root->ev_start, root->ev_stop));
  root->reduce_ms = ms;
  if (stats_out) {  // work counters summed over the ranks; times of the slowest rank
    *stats_out = sts[0];
    for (size_t i = 1; i < R; ++i) {
      const pt_stats& a = sts[i];
      stats_out->paths += a.paths; stats_out->rays += a.rays; stats_out->node_visits += a.node_visits;
      stats_out->tri_tests += a.tri_tests; stats_out->hits_shaded += a.hits_shaded; stats_out->texture_taps += a.texture_taps;
      stats_out->shaded_vertices += a.shaded_vertices; stats_out->nan_samples += a.nan_samples;
      stats_out->rays_closest += a.rays_closest; stats_out->rays_shadow_walked += a.rays_shadow_walked;
      stats_out->rays_shadow_shortcut += a.rays_shadow_shortcut;
      if (a.render_ms > stats_out->render_ms) stats_out->render_ms = a.render_ms;
      if (a.iterations > stats_out->iterations) stats_out->iterations = a.iterations;
    }
    stats_out->reduce_ms = ms;
  }
  return PT_OK;
}

}  // namespace

extern "C" int pt_render(pt_ctx* ctx, const pt_render_params* params, uint8_t* rgba8_out, float* accum_out,
                         uint32_t* primary_id_out, pt_stats* stats_out) {
  if (!ctx) return PT_ERR_INVALID;
  if (!params) return fail(ctx, PT_ERR_INVALID, "params is NULL");
  PT_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ctx->members.empty() ? render_accum(ctx, params, nullptr, nullptr, ctx->stream, stats_out) : render_group(ctx, params, stats_out);
  if (rc != PT_OK) return rc;
  const uint32_t pixels = params->pc.resolution_x * params->pc.resolution_y;
  PT_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  if (rgba8_out) {
    ptd::launch_resolve(ctx->accum_out, pixels, ctx->rgba8_out, ctx->stream);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(rgba8_out, ctx->rgba8_out, 4 * static_cast<size_t>(pixels), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (accum_out)
    PT_HIP(ctx, hipMemcpyAsync(accum_out, ctx->accum_out, 16 * static_cast<size_t>(pixels), hipMemcpyDeviceToHost, ctx->stream));
  if (primary_id_out)
    PT_HIP(ctx, hipMemcpyAsync(primary_id_out, ctx->primary_out, 4 * static_cast<size_t>(pixels), hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (stats_out) {
    float ms = 0.0f;
    PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    stats_out->download_ms = ms;
  }
  return PT_OK;
}
