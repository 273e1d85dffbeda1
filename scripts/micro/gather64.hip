// Microbenchmark: dependent random 64-byte record gathers (a BVH walk's access shape) from a table far larger
// than L2, with `active` of every 64 lanes walking.  A: each walking lane issues 4 x global_load_dwordx4 for its
// record.  B: the wave fetches the walking lanes' records cooperatively — 4 lanes x 16 bytes per record, one
// load instruction per 16 records — and hands them over through LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void walk_a(const float4* __restrict__ table, uint32_t records, uint32_t steps, uint32_t active, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t ref = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % records;
  float acc = 0.0f;
  if (lane < active) {
    for (uint32_t s = 0; s < steps; ++s) {
      const float4 a = table[4ull * ref], b = table[4ull * ref + 1], c = table[4ull * ref + 2], d = table[4ull * ref + 3];
      acc += a.x + b.y + c.z + d.x;
      ref = ((__float_as_uint(a.w) ^ __float_as_uint(c.w)) + (s + 1u) * 2246822519u + (blockIdx.x * blockDim.x + threadIdx.x) * 3266489917u) % records;  // dependent next record
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ref + __float_as_uint(acc);
}

// A': as A, but the `active` walking lanes are spread over the wave (lane % (64 / active) == 0): one lane per quad at 16
__global__ __launch_bounds__(256) void walk_s(const float4* __restrict__ table, uint32_t records, uint32_t steps, uint32_t active, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t ref = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % records;
  float acc = 0.0f;
  if (lane % (64u / active) == 0u) {
    for (uint32_t s = 0; s < steps; ++s) {
      const float4 a = table[4ull * ref], b = table[4ull * ref + 1], c = table[4ull * ref + 2], d = table[4ull * ref + 3];
      acc += a.x + b.y + c.z + d.x;
      ref = ((__float_as_uint(a.w) ^ __float_as_uint(c.w)) + (s + 1u) * 2246822519u + (blockIdx.x * blockDim.x + threadIdx.x) * 3266489917u) % records;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ref + __float_as_uint(acc);
}

__global__ __launch_bounds__(256) void walk_b(const float4* __restrict__ table, uint32_t records, uint32_t steps, uint32_t active, uint32_t* out) {
  __shared__ uint32_t refs[4][16];
  __shared__ float4 recs[4][16][4];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint32_t ref = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % records;
  float acc = 0.0f;
  const bool walking = lane < active;  // active <= 16 here
  for (uint32_t s = 0; s < steps; ++s) {
    if (walking) refs[w][lane] = ref;
    __builtin_amdgcn_wave_barrier();
    const uint32_t r = lane >> 2, part = lane & 3u;
    float4 v = make_float4(0, 0, 0, 0);
    if (r < active) v = table[4ull * refs[w][r] + part];
    if (r < active) recs[w][r][part] = v;
    __builtin_amdgcn_wave_barrier();
    if (walking) {
      const float4 a = recs[w][lane][0], b = recs[w][lane][1], c = recs[w][lane][2], d = recs[w][lane][3];
      acc += a.x + b.y + c.z + d.x;
      ref = ((__float_as_uint(a.w) ^ __float_as_uint(c.w)) + (s + 1u) * 2246822519u + (blockIdx.x * blockDim.x + threadIdx.x) * 3266489917u) % records;
    }
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ref + __float_as_uint(acc);
}

int main(int argc, char** argv) {
  const uint32_t records = argc > 1 ? atoi(argv[1]) : 8u << 20;  // x 64 B = 512 MiB
  const uint32_t steps = 256, blocks = 256 * 20;
  std::vector<uint32_t> host(16ull * records);
  uint32_t x = 12345;
  for (auto& v : host) { x = x * 1664525u + 1013904223u; v = x >> 4; }
  float4* table; uint32_t* out;
  CHECK(hipMalloc(&table, 64ull * records));
  CHECK(hipMalloc(&out, 4ull * blocks * 256));
  CHECK(hipMemcpy(table, host.data(), 64ull * records, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t active : {4u, 8u, 16u, 64u}) {
    for (int variant = 0; variant < 3; ++variant) {
      if (variant >= 1 && active > 16) continue;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (variant == 2) hipLaunchKernelGGL(walk_s, dim3(blocks), dim3(256), 0, 0, table, records, steps, active, out);
        else if (variant == 0) hipLaunchKernelGGL(walk_a, dim3(blocks), dim3(256), 0, 0, table, records, steps, active, out);
        else hipLaunchKernelGGL(walk_b, dim3(blocks), dim3(256), 0, 0, table, records, steps, active, out);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double fetches = double(blocks) * 4 * active * steps;
      printf("active %2u  %s  %8.3f ms  %7.2f Grecords/s  %7.1f GB/s\n", active, variant == 2 ? "scat" : (variant ? "coop" : "lane"), best, fetches / best / 1e6, fetches * 64 / best / 1e6);
    }
  }
  return 0;
}
