// pathtracer [config.json | list_devices] — the reference's CLI (src/main.cc:12-91) on the
// HIP renderer. Same config keys (resolution_x, resolution_y, sample_count, bg_color as
// [r,g,b] or "#RRGGBB", gltf_path, optional device_id); build-only optional keys: "seed"
// (the reference's seed is not reproducible, Q4), "stats" (true: print the work counters) and "num_gpus" (GPUs of this
// node that split every pixel's strata).
// Writes <gltf stem>.png into the current directory, like the reference.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "image_io.h"
#include "json.h"
#include "renderer.h"

namespace {

using pt::timeit;

std::string stem_of(const std::string& path) {  // std::filesystem::path(p).stem()
  const size_t slash = path.find_last_of("/\\");
  std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = name.find_last_of('.');
  if (dot != std::string::npos && dot != 0) name.resize(dot);
  return name;
}

}  // namespace

int main(int argc, char** argv) {
  try {
    if (argc == 2 && std::string(argv[1]) == "list_devices") {
      pt::list_devices();
      return 0;
    }
    std::string config_path = "config.json";
    if (argc == 2) config_path = argv[1];

    std::ifstream config_file(config_path);
    if (!config_file) throw std::runtime_error("Failed to open file: " + config_path);
    std::stringstream ss;
    ss << config_file.rdbuf();
    const ptjson::Value config = ptjson::parse(ss.str());

    pt::RenderConfig rc;
    rc.resolution_x = static_cast<uint32_t>(config.at("resolution_x").as_int());
    rc.resolution_y = static_cast<uint32_t>(config.at("resolution_y").as_int());
    rc.sample_count = static_cast<uint32_t>(config.at("sample_count").as_int());
    const ptjson::Value& bg = config.at("bg_color");
    if (bg.is_array()) {
      for (int k = 0; k < 3; ++k) rc.bg_color[k] = static_cast<float>(bg.at(static_cast<size_t>(k)).as_number());
    } else if (bg.is_string()) {  // "#RRGGBB" (main.cc:41-52)
      const unsigned long hex = std::strtoul(bg.as_string().substr(1).c_str(), nullptr, 16);
      rc.bg_color[0] = static_cast<float>((hex >> 16) & 0xFF) / 255.0f;
      rc.bg_color[1] = static_cast<float>((hex >> 8) & 0xFF) / 255.0f;
      rc.bg_color[2] = static_cast<float>(hex & 0xFF) / 255.0f;
    } else {
      throw std::runtime_error("Invalid bg_color format.");
    }
    uint32_t device_id = static_cast<uint32_t>(-1);
    if (config.contains("device_id")) device_id = static_cast<uint32_t>(config.at("device_id").as_int());

    // build-only key: "num_gpus" (> 0: that many GPUs from device_id on share every pixel's strata, one RCCL sum-reduce)
    const uint32_t num_gpus = config.contains("num_gpus") ? static_cast<uint32_t>(config.at("num_gpus").as_int()) : 0;
    std::unique_ptr<pt::Renderer> renderer_holder(num_gpus ? new pt::Renderer(device_id, num_gpus) : new pt::Renderer(device_id));
    pt::Renderer& renderer = *renderer_holder;
    if (config.contains("seed")) renderer.set_seed(static_cast<uint32_t>(config.at("seed").as_int()));

    const std::string gltf_path = config.at("gltf_path").as_string();
    pt::Scene scene;
    timeit("GLTF Parsing", [&] { scene = pt::load_gltf(gltf_path); });

    const std::pair<const unsigned char*, size_t> result = renderer.render(scene, rc);
    const unsigned char* image = result.first;

    timeit("Saving PNG", [&] {
      const std::string result_path = stem_of(gltf_path) + ".png";
      if (!pt::write_png_rgba8(result_path, rc.resolution_x, rc.resolution_y, image, rc.resolution_x * 4))
        throw std::runtime_error("Failed to write " + result_path);
    });

    if (config.contains("stats") && config.at("stats").boolean) {
      const pt_stats& st = renderer.stats();
      std::printf("{\"paths\": %llu, \"rays\": %llu, \"node_visits\": %llu, \"tri_tests\": %llu, \"render_ms\": %.3f, "
                  "\"mrays_per_s\": %.1f, \"mpaths_per_s\": %.1f, \"reduce_ms\": %.3f}\n",
                  static_cast<unsigned long long>(st.paths), static_cast<unsigned long long>(st.rays),
                  static_cast<unsigned long long>(st.node_visits), static_cast<unsigned long long>(st.tri_tests), st.render_ms,
                  st.render_ms > 0 ? static_cast<double>(st.rays) / st.render_ms / 1e3 : 0.0,
                  st.render_ms > 0 ? static_cast<double>(st.paths) / st.render_ms / 1e3 : 0.0, st.reduce_ms);
    }
  } catch (const std::exception& e) {
    std::printf("Exception Occured: %s\n", e.what());  // main.cc:86-88 (and exit code 0, as the reference)
  }
  return 0;
}
