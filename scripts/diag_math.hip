// Diagnostic: which primitive differs between host and gfx950 under the build flags?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
#include <cstring>
#include "pt_glsl_math.h"

#define NOPS 12
__host__ __device__ inline float op(int k, float a, float b) {
  float s, c;
  switch (k) {
    case 0: return a / b;
    case 1: return sqrtf(a);
    case 2: return 1.0f / __builtin_sqrtf(a);
    case 3: pt_sincos(a * 6.0f, &s, &c); return s;
    case 4: pt_sincos(a * 6.0f, &s, &c); return c;
    case 5: return pt_pow(a, 2.2f);
    case 6: return pt_pow(a, 5.0f);
    case 7: return pt_pow(a * 10.0f, (float)(1.0 / 2.2));
    case 8: return pt_exp2(a * 20.0f - 10.0f);
    case 9: return pt_log2_pos(a * 100.0f + 1e-3f);
    case 10: return a * b + a;
    case 11: return (float)(unsigned)(a * 4294967040.0f) / 4294967296.0f;
  }
  return 0;
}
__global__ void k(const float* a, const float* b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) for (int j = 0; j < NOPS; ++j) out[j * n + i] = op(j, a[i], b[i]);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> a(n), b(n), out(NOPS * n);
  srand(1);
  for (int i = 0; i < n; ++i) { a[i] = (rand() + 0.5f) / (float)RAND_MAX; b[i] = (rand() + 0.5f) / (float)RAND_MAX; }
  float *da, *db, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, NOPS * n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dout, n);
  hipMemcpy(out.data(), dout, NOPS * n * 4, hipMemcpyDeviceToHost);
  for (int j = 0; j < NOPS; ++j) {
    int bad = 0; int first = -1;
    for (int i = 0; i < n; ++i) { float h = op(j, a[i], b[i]); if (memcmp(&h, &out[j * n + i], 4)) { if (first < 0) first = i; ++bad; } }
    printf("op %d: %d mismatches", j, bad);
    if (first >= 0) printf(" first a=%a b=%a host=%a dev=%a", a[first], b[first], op(j, a[first], b[first]), out[j * n + first]);
    printf("\n");
  }
  return 0;
}
