// Host scene model: the reference's `struct Scene` (src/scene.h:8-85) with this build's
// C-ABI PODs as element types (byte-identical layouts, include/pathtracer_amd.h).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "pathtracer_amd.h"

namespace pt {

struct Image {  // Scene::Image, scene.h:47-50
  uint32_t width = 0, height = 0, component_count = 4;
  std::vector<unsigned char> data;  // RGBA8
};

struct Triangle {  // Scene::VertexIndices, scene.h:18
  uint32_t v[3];
};

struct Scene {  // scene.h:75-84
  std::vector<pt_vertex> vertices;
  std::vector<Triangle> triangles;
  std::vector<pt_material> materials;
  std::vector<Image> images;
  std::vector<pt_sampler> samplers;
  std::vector<pt_texture> textures;
  std::vector<pt_directional_light> directional_lights;
  std::vector<pt_point_light> point_lights;
  std::vector<uint32_t> emissive_triangle_indices;
  pt_camera camera{};
};

// load_gltf(path), src/gltf_loader/gltf_loader.h:5. Throws std::runtime_error.
Scene load_gltf(const std::string& path);

}  // namespace pt
