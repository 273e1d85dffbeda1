// Host-side data contract of the hot path: RenderJob::process_scene()
// (src/renderer/render_job.cc:76-121), the PushConstants constructor
// (src/renderer/render_job.cc:836-873) and initialize_work_group_size()
// (src/renderer/render_job.cc:63-74).  Compiled with -ffp-contract=off so the camera
// block matches the reference's (FMA-free) host arithmetic bit for bit.
#include <cmath>
#include <cstring>

#include "pathtracer_amd.h"

namespace {

struct F3 {
  float x, y, z;
};

inline F3 Load(const float* p) { return {p[0], p[1], p[2]}; }
inline void Store(float* p, F3 a) {
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
}
inline F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
inline F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline F3 operator/(F3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm::dot / glm::cross / glm::normalize (= v * inversesqrt(dot(v, v)))
inline float Dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 Cross(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline F3 Normalize(F3 a) {
  float inv = 1.0f / std::sqrt(Dot(a, a));
  return inv * a;
}

}  // namespace

extern "C" int pt_process_scene(const pt_vertex* vertices, uint32_t vertex_count, const uint32_t* triangles,
                                uint32_t triangle_count, const pt_material* materials,
                                uint32_t material_count, const pt_bvh_node* nodes, uint32_t node_count,
                                float* vertex_positions_out, pt_packed_vertex* packed_vertex_out,
                                uint32_t* emissive_out, uint32_t* emissive_count_out,
                                pt_packed_bvh_node* packed_nodes_out) {
  if (!vertices || !triangles || !materials || !nodes || !vertex_positions_out || !packed_vertex_out ||
      !emissive_out || !emissive_count_out || !packed_nodes_out)
    return PT_ERR_INVALID;

  const size_t corner_count = 3 * static_cast<size_t>(triangle_count);
  for (size_t c = 0; c < corner_count; ++c) {  // render_job.cc:80-97
    uint32_t vi = triangles[c];
    if (vi >= vertex_count) return PT_ERR_INVALID;
    const pt_vertex& v = vertices[vi];
    float* p = vertex_positions_out + 4 * c;
    p[0] = v.position[0];
    p[1] = v.position[1];
    p[2] = v.position[2];
    p[3] = 0.0f;
    pt_packed_vertex& o = packed_vertex_out[c];
    std::memcpy(o.position, p, sizeof(float) * 4);
    std::memcpy(o.normal_and_texcoord_u, v.normal, sizeof(float) * 3);
    o.normal_and_texcoord_u[3] = v.texcoord[0];
    std::memcpy(o.tangent_and_texcoord_v, v.tangent, sizeof(float) * 3);
    o.tangent_and_texcoord_v[3] = v.texcoord[1];
    std::memcpy(o.bitangent, v.bitangent, sizeof(float) * 3);
    o.material_index = v.material_index;
  }

  // Emissive list, computed on the PERMUTED triangle order from vertex 0's material (Q22).
  uint32_t emissive_n = 0;
  for (uint32_t t = 0; t < triangle_count; ++t) {  // render_job.cc:99-105
    int32_t mi = vertices[triangles[3 * static_cast<size_t>(t)]].material_index;
    if (mi < 0 || static_cast<uint32_t>(mi) >= material_count) return PT_ERR_INVALID;
    const pt_material& m = materials[mi];
    bool lit = m.emissive_factor[0] != 0.0f || m.emissive_factor[1] != 0.0f || m.emissive_factor[2] != 0.0f;
    if (lit || m.emissive_texture_index != -1) emissive_out[emissive_n++] = t;
  }
  *emissive_count_out = emissive_n;

  for (uint32_t i = 0; i < node_count; ++i) {  // render_job.cc:107-120
    const pt_bvh_node& n = nodes[i];
    pt_packed_bvh_node& o = packed_nodes_out[i];
    std::memcpy(o.aabb_min, n.aabb_min, sizeof(float) * 3);
    std::memcpy(o.aabb_max, n.aabb_max, sizeof(float) * 3);
    o.left_or_begin = n.triangle_count > 0 ? n.begin_index : n.left_child_index;
    o.triangle_count = n.triangle_count;
  }
  return PT_OK;
}

extern "C" int pt_make_push_constants(const pt_render_config* config, const pt_camera* camera,
                                      uint32_t emissive_triangle_count, uint32_t seed,
                                      pt_push_constants* out) {
  if (!config || !camera || !out) return PT_ERR_INVALID;
  if (config->resolution_x == 0 || config->resolution_y == 0) return PT_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  out->resolution_x = config->resolution_x;
  out->resolution_y = config->resolution_y;
  out->sample_count = config->sample_count;
  std::memcpy(out->bg_color, config->bg_color, sizeof(float) * 3);
  out->emissive_triangle_count = emissive_triangle_count;
  out->seed = seed;  // the reference: dist(rd) from std::random_device (render_job.cc:841-844, Q4)

  const float res_x = static_cast<float>(config->resolution_x);
  const float res_y = static_cast<float>(config->resolution_y);

  const F3 up{0.0f, 1.0f, 0.0f};
  const F3 w = Normalize(-Load(camera->lookat));  // lookat is a direction (Q18)
  const F3 u = Normalize(Cross(up, w));
  const F3 v = Cross(w, u);

  const float viewport_height = 2.0f;
  const float viewport_width = viewport_height * res_x / res_y;
  const float focal_length = 1.0f / std::tan(camera->vertical_fov / 2.0f);

  const F3 viewport_u = viewport_width * u;
  const F3 viewport_v = viewport_height * (-v);
  const F3 position = Load(camera->position);
  const F3 upper_left = position - focal_length * w - 0.5f * (viewport_u + viewport_v);

  Store(out->camera_position, position);
  Store(out->pixel_delta_u, viewport_u / res_x);
  Store(out->pixel_delta_v, viewport_v / res_y);
  Store(out->corner_pixel_pos, upper_left);  // the viewport's corner, no half-pixel offset
  return PT_OK;
}

extern "C" void pt_reference_work_group_size(uint32_t resolution_x, uint32_t resolution_y, uint32_t* local_x,
                                             uint32_t* local_y) {
  uint32_t lx = 16, ly = 32;
  if (resolution_x == resolution_y) lx = ly = 32;
  else if (resolution_x > resolution_y) lx = 32, ly = 16;
  if (local_x) *local_x = lx;
  if (local_y) *local_y = ly;
}
