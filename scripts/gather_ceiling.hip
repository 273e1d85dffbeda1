// What a wave64 kernel can gather from 64-byte records at random addresses on this chip: the ceiling the walk kernels' "algorithmic
// GB/s" should be read against. Each lane chases a chain of records: four 16-byte loads of one 64-byte record
// (exactly a pair-record visit), next index from the loaded data (DEPENDENT = 1, a traversal) or from the lane's own counter
// (DEPENDENT = 0, loads of successive trips overlap: the memory pipeline's throughput without the dependency).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/gather_ceiling.hip -o scripts/gather_ceiling.bin      Run: scripts/gather_ceiling.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // triple32-like
  x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
  return x;
}

template <bool DEPENDENT, int QUADS>
__global__ __launch_bounds__(256, 5) void k_gather(const float4* rec, uint32_t mask, uint32_t trips, uint32_t active_of_64, float* out) {
  const uint32_t lane = threadIdx.x & 63u;
  // `active_of_64` lanes of every wave work (spread over the wave as a walk's live lanes are), the rest idle
  const bool on = (lane * active_of_64) / 64u != ((lane + 1u) * active_of_64) / 64u || active_of_64 == 64u;
  uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 1u) & mask;
  uint32_t ctr = idx;
  float acc = 0.0f;
  if (on) {
    for (uint32_t t = 0; t < trips; ++t) {
      const float4* p = rec + 4u * static_cast<size_t>(idx);
      float4 q[4];
#pragma unroll
      for (int k = 0; k < QUADS; ++k) q[k] = p[k];
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < QUADS; ++k) s += q[k].x + q[k].w;
      acc += s;
      if (DEPENDENT) idx = (__float_as_uint(q[QUADS - 1].w) ^ (ctr = ctr * 1664525u + 1013904223u)) & mask;  // the record names the next one (random words), salted per lane: chains of one map would merge
      else idx = (ctr = mix(ctr + t)) & mask;
    }
  }
  if (acc == 123.456f) out[0] = acc;  // keeps the loads
}

int main(int argc, char** argv) {
  const uint32_t trips = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const uint32_t cus = prop.multiProcessorCount;
  printf("%s, %u CUs; %u trips per lane; 5 waves per SIMD (20 per CU), one 256-thread block per 4 waves\n", prop.name, cus, trips);
  const size_t max_records = 1u << 23;  // 512 MB
  std::vector<uint32_t> host(16 * max_records);
  uint32_t x = 12345u;
  for (auto& w : host) { x = x * 1664525u + 1013904223u; w = x >> 3; }
  float4* rec; float* out;
  CHECK(hipMalloc(&rec, 64 * max_records));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemcpy(rec, host.data(), 64 * max_records, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const uint32_t blocks = cus * 5;
  printf("%-12s %-10s %-7s %-6s %10s %10s %12s\n", "records", "footprint", "lanes", "chain", "ms", "GB/s", "Gvisits/s");
  for (uint32_t log_n : {12u, 16u, 18u, 20u, 23u}) {
    for (uint32_t active : {64u, 23u, 16u}) {
      for (int dep = 1; dep >= 0; --dep) {
        for (int quads : {4, 1}) {
          const uint32_t mask = (1u << log_n) - 1u;
          float ms = 0;
          for (int rep = 0; rep < 2; ++rep) {  // first repetition warms
            CHECK(hipEventRecord(e0));
            if (dep && quads == 4) k_gather<true, 4><<<blocks, 256>>>(rec, mask, trips, active, out);
            else if (dep) k_gather<true, 1><<<blocks, 256>>>(rec, mask, trips, active, out);
            else if (quads == 4) k_gather<false, 4><<<blocks, 256>>>(rec, mask, trips, active, out);
            else k_gather<false, 1><<<blocks, 256>>>(rec, mask, trips, active, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
          }
          const double visits = double(blocks) * 4 * active * trips;
          printf("2^%-10u %6.1f MB  %-7u %-6s %10.3f %10.0f %12.2f   (%d x 16 B per visit)\n", log_n, 64.0 * (mask + 1.0) / 1e6, active,
                 dep ? "dep" : "indep", ms, visits * 16.0 * quads / ms / 1e6, visits / ms / 1e6, quads);
        }
      }
    }
  }
  return 0;
}
