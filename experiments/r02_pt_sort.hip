// EXPERIMENT (PT_SORT_RAYS): re-order the wavefront pipeline's ray queue by (origin cell, direction octant) before
// every traversal launch, so that the 64 rays of a wave start in the same region and head the same way. State stays
// in its slot (k_extend reads ray_o/ray_d[slot], writes hit[slot]), so results do not change — only the order in
// which slots are traversed. Sorting is rocPRIM's radix sort here: this file decides whether the idea is worth a
// hand-written binning pass at all.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

#include "pt_device.h"

namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__device__ __forceinline__ uint32_t ray_key(float4 o, float4 d, float3 lo, float3 scale, uint32_t bits, uint32_t mode) {
  const float cells = static_cast<float>(1u << bits);
  auto cell = [&](float x, float l, float s) {
    float c = (x - l) * s * cells;
    c = c < 0.0f ? 0.0f : (c > cells - 1.0f ? cells - 1.0f : c);
    return static_cast<uint32_t>(c);
  };
  const uint32_t m = spread3(cell(o.x, lo.x, scale.x)) | (spread3(cell(o.y, lo.y, scale.y)) << 1) | (spread3(cell(o.z, lo.z, scale.z)) << 2);
  const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  if (mode == 2u) return (oct << (3u * bits)) | m;  // octant-major
  return (m << 3) | oct;                            // origin-major
}

__global__ __launch_bounds__(256) void k_ray_keys(pt_dev_state st, const uint32_t* queue, uint32_t counter_word, uint32_t bound, float3 lo,
                                                  float3 scale, uint32_t bits, uint32_t mode, uint32_t* keys, const float4* origin,
                                                  const float4* dir) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const uint32_t n = st.counters[counter_word];
  uint32_t key = 0xFFFFFFFFu;
  if (i < n) {
    const uint32_t slot = queue[i];
    key = ray_key(origin[slot], dir[slot], lo, scale, bits, mode);
  }
  keys[i] = key;
}

}  // namespace

namespace ptd {

size_t sort_temp_bytes(uint32_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 32, hipStream_t(nullptr));
  return bytes;
}

void launch_ray_keys(const pt_dev_state& st, const uint32_t* queue, uint32_t counter_word, uint32_t bound, const float* lo, const float* hi,
                     uint32_t bits, uint32_t mode, uint32_t* keys, const float4* origin, const float4* dir, hipStream_t s) {
  const float3 l = make_float3(lo[0], lo[1], lo[2]);
  const float3 sc = make_float3(1.0f / (hi[0] - lo[0] + 1e-20f), 1.0f / (hi[1] - lo[1] + 1e-20f), 1.0f / (hi[2] - lo[2] + 1e-20f));
  hipLaunchKernelGGL(k_ray_keys, dim3((bound + 255) / 256), dim3(256), 0, s, st, queue, counter_word, bound, l, sc, bits, mode, keys, origin, dir);
}

int sort_queue(void* temp, size_t temp_bytes, uint32_t* keys_in, uint32_t* keys_out, const uint32_t* q_in, uint32_t* q_out, uint32_t n,
               uint32_t key_bits, hipStream_t s) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, q_in, q_out, n, 0, key_bits, s) == hipSuccess ? 0 : -1;
}

}  // namespace ptd
