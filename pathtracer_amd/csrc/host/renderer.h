// C++ mirror of the reference's renderer interface (src/renderer/renderer.h:16-55) on top of
// the C ABI: same names, same argument meaning. A reference maintainer swaps this header for
// theirs and main.cc keeps compiling.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pathtracer_amd.h"
#include "scene.h"

#define NR_CHANNELS 4

namespace pt {

struct RenderConfig {  // renderer.h:16-20
  uint32_t resolution_x = 0, resolution_y = 0;
  uint32_t sample_count = 0;
  float bg_color[3] = {0, 0, 0};
};

class Renderer {
 public:
  explicit Renderer(uint32_t device_index = static_cast<uint32_t>(-1));  // renderer.h:24
  // build-only: num_gpus GPUs of this node, device_index, device_index + 1, ... (config.json key "num_gpus"): the strata of
  // every pixel are split over them and the accumulators sum-reduced with RCCL (pt_create_multi)
  Renderer(uint32_t device_index, uint32_t num_gpus);
  ~Renderer();
  Renderer(const Renderer&) = delete;
  Renderer& operator=(const Renderer&) = delete;

  // Renderer::render (renderer.h:26-27). Like the reference it permutes scene.triangles
  // (build_bvh) and appends scene.emissive_triangle_indices; unlike the reference the
  // returned pixels are owned by this object (valid until the next render), not by a
  // destroyed RenderJob.
  std::pair<const unsigned char*, size_t> render(Scene& scene, const RenderConfig& config);

  void set_seed(uint32_t seed) { seed_ = seed; has_seed_ = true; }  // build-only: the reference draws it from std::random_device
  const pt_stats& stats() const { return stats_; }

 private:
  pt_ctx* ctx_ = nullptr;
  std::vector<unsigned char> image_;
  pt_stats stats_{};
  uint32_t seed_ = 0;
  bool has_seed_ = false;
};

void list_devices();  // renderer.h:55

// timeit(), src/timeit.h:5-24 (same output format: "{:<16} -> " + ms / s); shared by the CLI and the Renderer binding
template <class F>
void timeit(const char* info, F&& f) {
  std::printf("%-16s -> ", info);
  std::fflush(stdout);
  const auto t1 = std::chrono::steady_clock::now();
  f();
  const auto t2 = std::chrono::steady_clock::now();
  const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
  if (ms < 100) std::printf("%lldms\n", ms);
  else if (ms < 10000) std::printf("%.2gs\n", static_cast<double>(ms) / 1000.0);
  else std::printf("%llds\n", ms / 1000);
}

}  // namespace pt
