#include "renderer.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>

namespace pt {


Renderer::Renderer(uint32_t device_index) {
  // the reference honours only ids > 0 and otherwise takes the first discrete GPU
  // (renderer.cc:58-70, Q21); here every id >= 0 is honoured and -1 means GPU 0
  ctx_ = pt_create(device_index == static_cast<uint32_t>(-1) ? -1 : static_cast<int>(device_index));
  if (!ctx_) throw std::runtime_error(std::string("Failed to create the renderer: ") + pt_last_error(nullptr));
}

Renderer::Renderer(uint32_t device_index, uint32_t num_gpus) {
  if (num_gpus == 0) num_gpus = 1;
  const int first = device_index == static_cast<uint32_t>(-1) ? 0 : static_cast<int>(device_index);
  std::vector<int> ids;
  for (uint32_t i = 0; i < num_gpus; ++i) ids.push_back(first + static_cast<int>(i));
  ctx_ = pt_create_multi(ids.data(), static_cast<int>(ids.size()));
  if (!ctx_) throw std::runtime_error(std::string("Failed to create the renderer: ") + pt_last_error(nullptr));
}

Renderer::~Renderer() { pt_destroy(ctx_); }

std::pair<const unsigned char*, size_t> Renderer::render(Scene& scene, const RenderConfig& config) {
  // RenderJob ctor, src/renderer/render_job.cc:20-61
  const uint32_t ntri = static_cast<uint32_t>(scene.triangles.size());
  std::vector<pt_bvh_node> nodes(2 * static_cast<size_t>(ntri));
  uint32_t node_count = 0, depth = 0;
  int rc = PT_OK;
  timeit("BVH Construction", [&] {
    rc = pt_build_bvh(scene.vertices.data(), static_cast<uint32_t>(scene.vertices.size()),
                      reinterpret_cast<uint32_t*>(scene.triangles.data()), ntri, 16, nodes.data(), &node_count, &depth);
  });
  if (rc != PT_OK) throw std::runtime_error("build_bvh failed (empty scene or vertex index out of range)");

  std::vector<float> vertex_positions(12 * static_cast<size_t>(ntri));
  std::vector<pt_packed_vertex> packed_vertices(3 * static_cast<size_t>(ntri));
  std::vector<pt_packed_bvh_node> packed_nodes(node_count);
  std::vector<uint32_t> emissive(ntri);
  uint32_t emissive_count = 0;
  rc = pt_process_scene(scene.vertices.data(), static_cast<uint32_t>(scene.vertices.size()),
                        reinterpret_cast<const uint32_t*>(scene.triangles.data()), ntri, scene.materials.data(),
                        static_cast<uint32_t>(scene.materials.size()), nodes.data(), node_count, vertex_positions.data(),
                        packed_vertices.data(), emissive.data(), &emissive_count, packed_nodes.data());
  if (rc != PT_OK) throw std::runtime_error("process_scene failed (material index out of range)");
  scene.emissive_triangle_indices.insert(scene.emissive_triangle_indices.end(), emissive.begin(), emissive.begin() + emissive_count);  // render_job.cc:103

  std::vector<pt_image> images;
  for (const Image& im : scene.images) images.push_back(pt_image{im.width, im.height, im.data.data()});

  pt_scene_desc d{};
  d.triangle_count = ntri;
  d.vertex_positions = vertex_positions.data();
  d.packed_vertex_data = packed_vertices.data();
  d.bvh_nodes = packed_nodes.data();
  d.bvh_node_count = node_count;
  d.bvh_max_depth = depth;
  d.emissive_triangles = scene.emissive_triangle_indices.data();
  d.emissive_triangle_count = static_cast<uint32_t>(scene.emissive_triangle_indices.size());
  d.materials = scene.materials.data();
  d.material_count = static_cast<uint32_t>(scene.materials.size());
  d.directional_lights = scene.directional_lights.data();
  d.directional_light_count = scene.directional_lights.empty() ? 0 : static_cast<uint32_t>(scene.directional_lights.size() - 1);  // render_job.cc:567
  d.point_lights = scene.point_lights.data();
  d.point_light_count = scene.point_lights.empty() ? 0 : static_cast<uint32_t>(scene.point_lights.size() - 1);  // :568
  d.textures = scene.textures.data();
  d.texture_count = static_cast<uint32_t>(scene.textures.size());
  d.images = images.data();
  d.image_count = static_cast<uint32_t>(images.size());
  d.samplers = scene.samplers.data();
  d.sampler_count = static_cast<uint32_t>(scene.samplers.size());
  if (pt_upload_scene(ctx_, &d) != PT_OK) throw std::runtime_error(std::string("scene upload failed: ") + pt_last_error(ctx_));

  pt_render_params params{};
  pt_render_config cfg{config.resolution_x, config.resolution_y, config.sample_count,
                       {config.bg_color[0], config.bg_color[1], config.bg_color[2]}};
  uint32_t seed = seed_;
  if (!has_seed_) {  // PushConstants ctor: seed = dist(rd) (render_job.cc:841-844)
    std::random_device rd;
    seed = static_cast<uint32_t>(rd());
  }
  if (pt_make_push_constants(&cfg, &scene.camera, d.emissive_triangle_count, seed, &params.pc) != PT_OK)
    throw std::runtime_error("invalid render configuration");

  image_.assign(static_cast<size_t>(config.resolution_x) * config.resolution_y * NR_CHANNELS, 0);
  timeit("Rendering", [&] { rc = pt_render(ctx_, &params, image_.data(), nullptr, nullptr, &stats_); });
  if (rc != PT_OK) throw std::runtime_error(std::string("render failed: ") + pt_last_error(ctx_));
  return {image_.data(), image_.size()};
}

void list_devices() {  // renderer.cc:196-213: "{index}: {name}"
  const int n = pt_device_count();
  for (int i = 0; i < n; ++i) {
    char name[256];
    if (pt_device_name(i, name, sizeof(name)) == PT_OK) std::printf("%d: %s\n", i, name);
  }
}

}  // namespace pt
