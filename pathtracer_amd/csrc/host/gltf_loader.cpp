// glTF 2.0 / GLB front end producing the reference's `Scene`
// (src/gltf_loader/gltf_loader.cc:93-494), without tinygltf / glm / stb: own JSON reader,
// GLB container parsing, accessor decoding and PNG decode. What is kept from the reference
// is its BEHAVIOUR — the index conventions (material 0 / image 0 / sampler 0 are built-in
// defaults, texture indices are not shifted, one dummy texture if none, one zero light
// appended to each light array: SURVEY.md Q17), the light intensity scale 4*pi/683 (Q9), the
// auto-camera (Q18), u16/u32 indices, normal-matrix transforms. The matrix helpers restate
// glm's formulas (translate/scale/mat4_cast/inverse) in float; glm itself is not available
// here, so their last-bit agreement with glm is unpinned.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

#include "image_io.h"
#include "json.h"
#include "scene.h"

namespace pt {

namespace {

using ptjson::Value;

// ------------------------------------------------------------------ glm-like float math
struct V3 {
  float x = 0, y = 0, z = 0;
};
struct V4 {
  float x = 0, y = 0, z = 0, w = 0;
};
struct M4 {  // column-major: c[col]
  V4 c[4];
};

V4 operator*(const V4& a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
V4 operator+(const V4& a, const V4& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

M4 identity() {
  M4 m;
  m.c[0] = {1, 0, 0, 0};
  m.c[1] = {0, 1, 0, 0};
  m.c[2] = {0, 0, 1, 0};
  m.c[3] = {0, 0, 0, 1};
  return m;
}

V4 mul(const M4& m, const V4& v) {  // glm: m[0]*v.x + m[1]*v.y + (m[2]*v.z + m[3]*v.w) pattern
  const V4 a = m.c[0] * v.x, b = m.c[1] * v.y, c = m.c[2] * v.z, d = m.c[3] * v.w;
  return (a + b) + (c + d);
}

M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int j = 0; j < 4; ++j) {
    const V4& bj = b.c[j];
    r.c[j] = ((a.c[0] * bj.x + a.c[1] * bj.y) + a.c[2] * bj.z) + a.c[3] * bj.w;
  }
  return r;
}

M4 translate(const M4& m, V3 v) {  // glm::translate
  M4 r = m;
  r.c[3] = ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3];
  return r;
}

M4 scale(const M4& m, V3 v) {  // glm::scale
  M4 r;
  r.c[0] = m.c[0] * v.x;
  r.c[1] = m.c[1] * v.y;
  r.c[2] = m.c[2] * v.z;
  r.c[3] = m.c[3];
  return r;
}

M4 quat_to_mat4(float w, float x, float y, float z) {  // glm::mat4_cast
  const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y,
              qwz = w * z;
  M4 r = identity();
  r.c[0] = {1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy), 0};
  r.c[1] = {2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx), 0};
  r.c[2] = {2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy), 0};
  return r;
}

float at(const M4& m, int col, int row) { return (&m.c[col].x)[row]; }

M4 inverse(const M4& m) {  // cofactor expansion (glm::inverse's algorithm)
  const float c00 = at(m, 2, 2) * at(m, 3, 3) - at(m, 3, 2) * at(m, 2, 3);
  const float c02 = at(m, 1, 2) * at(m, 3, 3) - at(m, 3, 2) * at(m, 1, 3);
  const float c03 = at(m, 1, 2) * at(m, 2, 3) - at(m, 2, 2) * at(m, 1, 3);
  const float c04 = at(m, 2, 1) * at(m, 3, 3) - at(m, 3, 1) * at(m, 2, 3);
  const float c06 = at(m, 1, 1) * at(m, 3, 3) - at(m, 3, 1) * at(m, 1, 3);
  const float c07 = at(m, 1, 1) * at(m, 2, 3) - at(m, 2, 1) * at(m, 1, 3);
  const float c08 = at(m, 2, 1) * at(m, 3, 2) - at(m, 3, 1) * at(m, 2, 2);
  const float c10 = at(m, 1, 1) * at(m, 3, 2) - at(m, 3, 1) * at(m, 1, 2);
  const float c11 = at(m, 1, 1) * at(m, 2, 2) - at(m, 2, 1) * at(m, 1, 2);
  const float c12 = at(m, 2, 0) * at(m, 3, 3) - at(m, 3, 0) * at(m, 2, 3);
  const float c14 = at(m, 1, 0) * at(m, 3, 3) - at(m, 3, 0) * at(m, 1, 3);
  const float c15 = at(m, 1, 0) * at(m, 2, 3) - at(m, 2, 0) * at(m, 1, 3);
  const float c16 = at(m, 2, 0) * at(m, 3, 2) - at(m, 3, 0) * at(m, 2, 2);
  const float c18 = at(m, 1, 0) * at(m, 3, 2) - at(m, 3, 0) * at(m, 1, 2);
  const float c19 = at(m, 1, 0) * at(m, 2, 2) - at(m, 2, 0) * at(m, 1, 2);
  const float c20 = at(m, 2, 0) * at(m, 3, 1) - at(m, 3, 0) * at(m, 2, 1);
  const float c22 = at(m, 1, 0) * at(m, 3, 1) - at(m, 3, 0) * at(m, 1, 1);
  const float c23 = at(m, 1, 0) * at(m, 2, 1) - at(m, 2, 0) * at(m, 1, 1);

  const V4 f0{c00, c00, c02, c03}, f1{c04, c04, c06, c07}, f2{c08, c08, c10, c11};
  const V4 f3{c12, c12, c14, c15}, f4{c16, c16, c18, c19}, f5{c20, c20, c22, c23};
  const V4 v0{at(m, 1, 0), at(m, 0, 0), at(m, 0, 0), at(m, 0, 0)};
  const V4 v1{at(m, 1, 1), at(m, 0, 1), at(m, 0, 1), at(m, 0, 1)};
  const V4 v2{at(m, 1, 2), at(m, 0, 2), at(m, 0, 2), at(m, 0, 2)};
  const V4 v3{at(m, 1, 3), at(m, 0, 3), at(m, 0, 3), at(m, 0, 3)};
  auto cw = [](const V4& a, const V4& b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; };
  auto sub = [](const V4& a, const V4& b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; };
  const V4 i0 = sub(cw(v1, f0), cw(v2, f1)) + cw(v3, f2);
  const V4 i1 = sub(cw(v0, f0), cw(v2, f3)) + cw(v3, f4);
  const V4 i2 = sub(cw(v0, f1), cw(v1, f3)) + cw(v3, f5);
  const V4 i3 = sub(cw(v0, f2), cw(v1, f4)) + cw(v2, f5);
  const V4 sa{+1, -1, +1, -1}, sb{-1, +1, -1, +1};
  M4 inv;
  inv.c[0] = cw(i0, sa);
  inv.c[1] = cw(i1, sb);
  inv.c[2] = cw(i2, sa);
  inv.c[3] = cw(i3, sb);
  const V4 row0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
  const V4 d = cw(m.c[0], row0);
  const float det = (d.x + d.y) + (d.z + d.w);
  const float one_over_det = 1.0f / det;
  for (int j = 0; j < 4; ++j) inv.c[j] = inv.c[j] * one_over_det;
  return inv;
}

struct M3 {
  V3 c[3];
};

M3 normal_matrix(const M4& m) {  // mat3(transpose(inverse(m)))
  const M4 inv = inverse(m);
  M3 r;
  for (int col = 0; col < 3; ++col) r.c[col] = {at(inv, 0, col), at(inv, 1, col), at(inv, 2, col)};
  return r;
}

V3 mul(const M3& m, V3 v) {
  return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z, m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
          m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}

V3 normalize(V3 v) {
  const float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  return {v.x * inv, v.y * inv, v.z * inv};
}

V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// ------------------------------------------------------------------------------ loader
std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Failed to open file: " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

std::string dir_of(const std::string& path) {
  const size_t p = path.find_last_of("/\\");
  return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

bool ends_with(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

class Loader {
 public:
  Loader(const std::string& path, Scene& scene) : scene_(scene), base_dir_(dir_of(path)) {
    const std::string bytes = read_file(path);
    std::string json_text;
    if (ends_with(path, ".glb")) {  // gltf_loader.cc:126-130: the extension decides
      if (bytes.size() < 20 || bytes.compare(0, 4, "glTF") != 0) throw std::runtime_error("Failed to parse glTF\n");
      auto u32 = [&](size_t off) {
        uint32_t v;
        std::memcpy(&v, bytes.data() + off, 4);
        return v;
      };
      size_t pos = 12;
      const size_t total = std::min<size_t>(u32(8), bytes.size());
      while (pos + 8 <= total) {
        const uint32_t len = u32(pos), type = u32(pos + 4);
        if (pos + 8 + len > total) throw std::runtime_error("Failed to parse glTF\n");
        if (type == 0x4E4F534Au) json_text.assign(bytes, pos + 8, len);
        else if (type == 0x004E4942u && glb_bin_.empty()) glb_bin_.assign(bytes.begin() + pos + 8, bytes.begin() + pos + 8 + len);
        pos += 8 + ((len + 3u) & ~3u);
      }
      if (json_text.empty()) throw std::runtime_error("Failed to parse glTF\n");
    } else {
      json_text = bytes;
    }
    doc_ = ptjson::parse(json_text);
    load_buffers();
  }

  // Loader::process, gltf_loader.cc:145-182
  void process() {
    const Value* scenes = doc_.find("scenes");
    if (!scenes || scenes->size() == 0) throw std::runtime_error("glTF has no scene");
    int64_t scene_index = doc_.int_or("scene", 0);
    if (scene_index < 0 || static_cast<size_t>(scene_index) >= scenes->size()) scene_index = 0;
    camera_transform_ = identity();
    if (const Value* roots = scenes->at(static_cast<size_t>(scene_index)).find("nodes"))
      for (const Value& n : roots->array) process_node(node(n.as_int()), identity(), 0);

    scene_.materials.push_back(default_material());
    if (const Value* mats = doc_.find("materials"))
      for (const Value& m : mats->array) process_material(m);

    scene_.images.push_back(default_image());
    if (const Value* imgs = doc_.find("images"))
      for (const Value& im : imgs->array) process_image(im);

    scene_.samplers.push_back(pt_sampler{PT_FILTER_LINEAR, PT_FILTER_LINEAR, PT_WRAP_REPEAT, PT_WRAP_REPEAT});
    if (const Value* smps = doc_.find("samplers"))
      for (const Value& s : smps->array) process_sampler(s);

    if (const Value* texs = doc_.find("textures"))
      for (const Value& t : texs->array)  // indices shifted by one; texture indices themselves are not (Q17)
        scene_.textures.push_back(pt_texture{static_cast<int32_t>(t.int_or("sampler", -1)) + 1,
                                             static_cast<int32_t>(t.int_or("source", -1)) + 1});
    if (scene_.textures.empty()) scene_.textures.push_back(pt_texture{0, 0});

    scene_.directional_lights.push_back(pt_directional_light{});  // the trailing dummies (gltf_loader.cc:178-179)
    scene_.point_lights.push_back(pt_point_light{});

    process_camera();
  }

 private:
  static pt_material default_material() {  // gltf_loader.cc:12-22
    pt_material m{};
    m.base_color_factor[0] = m.base_color_factor[1] = m.base_color_factor[2] = 0.8f;
    m.base_color_factor[3] = 1.0f;
    m.base_color_texture_index = -1;
    m.metallic_factor = 0.0f;
    m.roughness_factor = 1.0f;
    m.metallic_roughness_texture_index = -1;
    m.occlusion_strength = 0.0f;
    m.occlusion_texture_index = -1;
    m.emissive_texture_index = -1;
    return m;
  }

  static Image default_image() {  // gltf_loader.cc:31-53: 64x64 checker
    Image im;
    im.width = im.height = 64;
    im.data.assign(64 * 64 * 4, 255);
    for (uint32_t i = 0; i < 64; ++i)
      for (uint32_t j = 0; j < 64; ++j)
        if ((i + j) % 2 == 0) {
          unsigned char* p = im.data.data() + 4 * (i * 64 + j);
          p[0] = p[1] = p[2] = 0;
        }
    return im;
  }

  const Value& node(int64_t i) const {
    const Value& nodes = doc_.at("nodes");
    if (i < 0 || static_cast<size_t>(i) >= nodes.size()) throw std::runtime_error("glTF: node index out of range");
    return nodes.array[static_cast<size_t>(i)];
  }

  std::vector<unsigned char> load_uri(const std::string& uri) const {
    if (uri.compare(0, 5, "data:") == 0) {
      const size_t comma = uri.find(',');
      if (comma == std::string::npos) throw std::runtime_error("glTF: bad data URI");
      return base64_decode(uri.substr(comma + 1));
    }
    // tinygltf percent-decodes external URIs before opening them ("my%20texture.png", as exporters write names with spaces)
    std::string decoded;
    for (size_t i = 0; i < uri.size(); ++i) {
      auto hex = [](char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1)); };
      if (uri[i] == '%' && i + 2 < uri.size() && hex(uri[i + 1]) >= 0 && hex(uri[i + 2]) >= 0) {
        decoded.push_back(static_cast<char>(hex(uri[i + 1]) * 16 + hex(uri[i + 2])));
        i += 2;
      } else {
        decoded.push_back(uri[i]);
      }
    }
    const std::string bytes = read_file(base_dir_ + decoded);
    return std::vector<unsigned char>(bytes.begin(), bytes.end());
  }

  void load_buffers() {
    const Value* bufs = doc_.find("buffers");
    if (!bufs) return;
    for (size_t i = 0; i < bufs->size(); ++i) {
      const Value& b = bufs->array[i];
      if (const Value* uri = b.find("uri")) buffers_.push_back(load_uri(uri->as_string()));
      else if (i == 0) buffers_.push_back(glb_bin_);
      else buffers_.emplace_back();
    }
  }

  struct View {
    const unsigned char* ptr = nullptr;
    size_t stride = 0, count = 0;
    int component_type = 0;
    std::string type;
  };

  // AccessorHelper, gltf_loader.cc:57-91
  View accessor(int64_t index, size_t element_size) const {
    const Value& accs = doc_.at("accessors");
    if (index < 0 || static_cast<size_t>(index) >= accs.size()) throw std::runtime_error("glTF: accessor index out of range");
    const Value& a = accs.array[static_cast<size_t>(index)];
    if (a.contains("sparse")) throw std::runtime_error("glTF: sparse accessors are not supported");
    const int64_t bv_index = a.int_or("bufferView", -1);
    const Value& bvs = doc_.at("bufferViews");
    if (bv_index < 0 || static_cast<size_t>(bv_index) >= bvs.size()) throw std::runtime_error("glTF: accessor without a buffer view");
    const Value& bv = bvs.array[static_cast<size_t>(bv_index)];
    const int64_t buf = bv.int_or("buffer", 0);
    if (buf < 0 || static_cast<size_t>(buf) >= buffers_.size()) throw std::runtime_error("glTF: buffer index out of range");
    View v;
    // counts and offsets come from the file: negative values are rejected before any cast, and the range check is written
    // so that nothing in it can wrap (count - 1 <= (size - offset - element_size) / stride)
    const int64_t count = a.int_or("count", 0), byte_stride = bv.int_or("byteStride", 0);
    const int64_t off_a = a.int_or("byteOffset", 0), off_v = bv.int_or("byteOffset", 0);
    if (count < 0 || byte_stride < 0 || off_a < 0 || off_v < 0) throw std::runtime_error("glTF: negative count / offset / stride in an accessor");
    v.count = static_cast<size_t>(count);
    v.component_type = static_cast<int>(a.int_or("componentType", 0));
    v.type = a.at("type").as_string();
    const size_t stride = static_cast<size_t>(byte_stride);
    v.stride = stride ? stride : element_size;
    const std::vector<unsigned char>& data = buffers_[static_cast<size_t>(buf)];
    const size_t size = data.size();
    if (static_cast<uint64_t>(off_a) > size || static_cast<uint64_t>(off_v) > size - static_cast<size_t>(off_a))
      throw std::runtime_error("glTF: accessor reads past the end of its buffer");
    const size_t offset = static_cast<size_t>(off_a) + static_cast<size_t>(off_v);
    if (v.count && (element_size > size - offset || (v.count - 1) > (size - offset - element_size) / v.stride))
      throw std::runtime_error("glTF: accessor reads past the end of its buffer");
    v.ptr = data.data() + offset;
    return v;
  }

  View float_attribute(const Value& attributes, const char* name, const char* type, size_t components) const {
    const Value* idx = attributes.find(name);
    if (!idx) return View{};
    View v = accessor(idx->as_int(), components * sizeof(float));
    if (v.component_type != 5126 || v.type != type)  // the reference asserts FLOAT + the exact type (gltf_loader.cc:355-365)
      throw std::runtime_error(std::string("glTF: attribute ") + name + " must be float " + type);
    return v;
  }

  // Loader::process_node, gltf_loader.cc:184-232
  void process_node(const Value& n, M4 transform, int depth) {
    if (depth > 512) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
    const Value* matrix = n.find("matrix");
    if (matrix && matrix->size() == 16) {
      M4 local;
      for (int i = 0; i < 16; ++i) (&local.c[i / 4].x)[i % 4] = static_cast<float>(matrix->array[static_cast<size_t>(i)].as_number());
      transform = mul(transform, local);
    } else {
      if (const Value* t = n.find("translation"); t && t->size() == 3)
        transform = translate(transform, V3{static_cast<float>(t->array[0].as_number()), static_cast<float>(t->array[1].as_number()),
                                            static_cast<float>(t->array[2].as_number())});
      if (const Value* r = n.find("rotation"); r && r->size() == 4)
        transform = mul(transform, quat_to_mat4(static_cast<float>(r->array[3].as_number()), static_cast<float>(r->array[0].as_number()),
                                                static_cast<float>(r->array[1].as_number()), static_cast<float>(r->array[2].as_number())));
      if (const Value* s = n.find("scale"); s && s->size() == 3)
        transform = scale(transform, V3{static_cast<float>(s->array[0].as_number()), static_cast<float>(s->array[1].as_number()),
                                        static_cast<float>(s->array[2].as_number())});
    }

    const int64_t mesh = n.int_or("mesh", -1);
    if (mesh != -1) {
      const Value& meshes = doc_.at("meshes");
      if (mesh < 0 || static_cast<size_t>(mesh) >= meshes.size()) throw std::runtime_error("glTF: mesh index out of range");
      for (const Value& prim : meshes.array[static_cast<size_t>(mesh)].at("primitives").array)
        if (prim.int_or("mode", 4) == 4) process_primitive(prim, transform);  // TRIANGLES only
    }

    if (const Value* ext = n.find("extensions"))
      if (const Value* khr = ext->find("KHR_lights_punctual")) {
        const int64_t li = khr->int_or("light", -1);
        const Value* lights = nullptr;
        if (const Value* dext = doc_.find("extensions"))
          if (const Value* dk = dext->find("KHR_lights_punctual")) lights = dk->find("lights");
        if (lights && li >= 0 && static_cast<size_t>(li) < lights->size()) process_light(lights->array[static_cast<size_t>(li)], transform);
      }

    if (n.int_or("camera", -1) == 0) camera_transform_ = transform;  // default_camera_index = 0

    if (const Value* children = n.find("children"))
      for (const Value& c : children->array) process_node(node(c.as_int()), transform, depth + 1);
  }

  // Loader::process_primitive + process_indices, gltf_loader.cc:369-426
  void process_primitive(const Value& prim, const M4& transform) {
    const Value& attributes = prim.at("attributes");
    const View positions = float_attribute(attributes, "POSITION", "VEC3", 3);
    const View normals = float_attribute(attributes, "NORMAL", "VEC3", 3);
    const View texcoords = float_attribute(attributes, "TEXCOORD_0", "VEC2", 2);
    const View tangents = float_attribute(attributes, "TANGENT", "VEC4", 4);
    if (!positions.ptr || !normals.ptr) throw std::runtime_error("glTF: primitive needs POSITION and NORMAL (the reference asserts both)");
    if (normals.count != positions.count || (texcoords.ptr && texcoords.count != positions.count))
      throw std::runtime_error("glTF: attribute counts differ");

    const size_t vertex_count = positions.count;
    const M3 nm = normal_matrix(transform);
    const int32_t material_index = static_cast<int32_t>(prim.int_or("material", -1)) + 1;
    auto load3 = [](const View& v, size_t i) {
      V3 r;
      std::memcpy(&r, v.ptr + i * v.stride, 12);
      return r;
    };
    for (size_t i = 0; i < vertex_count; ++i) {
      pt_vertex vert{};
      const V3 p = load3(positions, i), nrm = load3(normals, i);
      const V4 wp = mul(transform, V4{p.x, p.y, p.z, 1.0f});
      const V3 wn = normalize(mul(nm, nrm));
      vert.position[0] = wp.x; vert.position[1] = wp.y; vert.position[2] = wp.z;
      vert.normal[0] = wn.x; vert.normal[1] = wn.y; vert.normal[2] = wn.z;
      if (texcoords.ptr) std::memcpy(vert.texcoord, texcoords.ptr + i * texcoords.stride, 8);
      vert.material_index = material_index;
      if (tangents.ptr) {
        V4 t;
        std::memcpy(&t, tangents.ptr + i * tangents.stride, 16);
        V3 tan{t.x, t.y, t.z};
        V3 bit = cross(nrm, tan);
        bit = {bit.x * t.w, bit.y * t.w, bit.z * t.w};
        tan = normalize(mul(nm, tan));
        bit = normalize(mul(nm, bit));
        vert.tangent[0] = tan.x; vert.tangent[1] = tan.y; vert.tangent[2] = tan.z;
        vert.bitangent[0] = bit.x; vert.bitangent[1] = bit.y; vert.bitangent[2] = bit.z;
      }
      scene_.vertices.push_back(vert);
    }

    const int64_t indices = prim.int_or("indices", -1);
    if (indices < 0) throw std::runtime_error("glTF: non-indexed primitives are not supported");
    View iv = accessor(indices, 1);
    size_t isize;
    if (iv.component_type == 5123) isize = 2;
    else if (iv.component_type == 5125) isize = 4;
    else throw std::runtime_error("Unknown index type in gltf file.");  // gltf_loader.cc:422
    iv = accessor(indices, isize);
    auto index_at = [&](size_t i) -> uint32_t {
      if (isize == 2) {
        uint16_t v;
        std::memcpy(&v, iv.ptr + i * iv.stride, 2);
        return v;
      }
      uint32_t v;
      std::memcpy(&v, iv.ptr + i * iv.stride, 4);
      return v;
    };
    for (size_t i = 0; i + 2 < iv.count; i += 3) {
      Triangle t;
      for (int k = 0; k < 3; ++k) {
        const uint32_t local = index_at(i + static_cast<size_t>(k));
        if (local >= vertex_count) throw std::runtime_error("glTF: vertex index out of range");
        t.v[k] = index_offset_ + local;
      }
      scene_.triangles.push_back(t);
    }
    index_offset_ += static_cast<uint32_t>(vertex_count);
  }

  static int32_t texture_index(const Value& holder, const char* key) {
    const Value* t = holder.find(key);
    if (!t) return -1;
    if (t->int_or("texCoord", 0) != 0) throw std::runtime_error("glTF: only TEXCOORD_0 is supported (the reference asserts it)");
    return static_cast<int32_t>(t->int_or("index", -1));
  }

  // Loader::process_material, gltf_loader.cc:234-277
  void process_material(const Value& gm) {
    pt_material m{};
    static const Value kEmpty = [] { Value v; v.kind = Value::Kind::Object; return v; }();
    const Value* pbr_ptr = gm.find("pbrMetallicRoughness");
    const Value& pbr = pbr_ptr ? *pbr_ptr : kEmpty;
    const Value* bcf = pbr.find("baseColorFactor");
    for (int k = 0; k < 4; ++k) m.base_color_factor[k] = (bcf && bcf->size() == 4) ? static_cast<float>(bcf->array[static_cast<size_t>(k)].as_number()) : 1.0f;
    m.base_color_texture_index = texture_index(pbr, "baseColorTexture");
    m.metallic_factor = static_cast<float>(pbr.number_or("metallicFactor", 1.0));
    m.roughness_factor = static_cast<float>(pbr.number_or("roughnessFactor", 1.0));
    m.metallic_roughness_texture_index = texture_index(pbr, "metallicRoughnessTexture");
    const Value* nt = gm.find("normalTexture");
    m.normal_scale = static_cast<float>(nt ? nt->number_or("scale", 1.0) : 1.0);
    m.normal_texture_index = texture_index(gm, "normalTexture");
    const Value* ot = gm.find("occlusionTexture");
    m.occlusion_strength = static_cast<float>(ot ? ot->number_or("strength", 1.0) : 1.0);
    m.occlusion_texture_index = texture_index(gm, "occlusionTexture");
    const Value* ef = gm.find("emissiveFactor");
    for (int k = 0; k < 3; ++k) m.emissive_factor[k] = (ef && ef->size() == 3) ? static_cast<float>(ef->array[static_cast<size_t>(k)].as_number()) : 0.0f;
    m.emissive_texture_index = texture_index(gm, "emissiveTexture");
    if (const Value* ext = gm.find("extensions"))
      if (const Value* es = ext->find("KHR_materials_emissive_strength")) {
        const float strength = static_cast<float>(es->number_or("emissiveStrength", 1.0));
        for (int k = 0; k < 3; ++k) m.emissive_factor[k] *= strength;
      }
    scene_.materials.push_back(m);
  }

  // Loader::process_image, gltf_loader.cc:279-291 (tinygltf decodes to 4 x 8-bit)
  void process_image(const Value& gi) {
    std::vector<unsigned char> bytes;
    if (const Value* uri = gi.find("uri")) {
      bytes = load_uri(uri->as_string());
    } else {
      const int64_t bv_index = gi.int_or("bufferView", -1);
      const Value& bvs = doc_.at("bufferViews");
      if (bv_index < 0 || static_cast<size_t>(bv_index) >= bvs.size()) throw std::runtime_error("glTF: image without data");
      const Value& bv = bvs.array[static_cast<size_t>(bv_index)];
      const int64_t buf = bv.int_or("buffer", 0);
      const size_t off = static_cast<size_t>(bv.int_or("byteOffset", 0)), len = static_cast<size_t>(bv.int_or("byteLength", 0));
      if (buf < 0 || static_cast<size_t>(buf) >= buffers_.size() || off + len > buffers_[static_cast<size_t>(buf)].size())
        throw std::runtime_error("glTF: image buffer view out of range");
      bytes.assign(buffers_[static_cast<size_t>(buf)].begin() + static_cast<long>(off), buffers_[static_cast<size_t>(buf)].begin() + static_cast<long>(off + len));
    }
    Image im;
    decode_image_rgba8(bytes.data(), bytes.size(), im.width, im.height, im.data);
    scene_.images.push_back(std::move(im));
  }

  // Loader::process_sampler, gltf_loader.cc:293-335
  void process_sampler(const Value& gs) {
    auto filter = [](int64_t f) {
      switch (f) {
        case 9728: case 9984: case 9986: return PT_FILTER_NEAREST;  // NEAREST, NEAREST_MIPMAP_*
        default: return PT_FILTER_LINEAR;
      }
    };
    auto wrap = [](int64_t w) {
      switch (w) {
        case 33648: return PT_WRAP_MIRRORED_REPEAT;
        case 33071: return PT_WRAP_CLAMP_TO_EDGE;
        default: return PT_WRAP_REPEAT;
      }
    };
    scene_.samplers.push_back(pt_sampler{filter(gs.int_or("magFilter", -1)), filter(gs.int_or("minFilter", -1)),
                                         wrap(gs.int_or("wrapS", 10497)), wrap(gs.int_or("wrapT", 10497))});
  }

  // Loader::process_light, gltf_loader.cc:428-455
  void process_light(const Value& gl, const M4& transform) {
    float color[3] = {1.0f, 1.0f, 1.0f};
    if (const Value* c = gl.find("color"); c && c->size() == 3)
      for (int k = 0; k < 3; ++k) color[k] = static_cast<float>(c->array[static_cast<size_t>(k)].as_number());
    // float(intensity) * (4.0 * pi<float>() / 683.0): the factor is evaluated in double (Q9)
    const float intensity = static_cast<float>(static_cast<double>(static_cast<float>(gl.number_or("intensity", 1.0))) *
                                               (4.0 * static_cast<double>(3.14159265358979323846264338327950288f) / 683.0));
    const std::string type = gl.contains("type") ? gl.at("type").as_string() : std::string();
    if (type == "directional") {
      pt_directional_light l{};
      std::memcpy(l.color, color, 12);
      l.intensity = intensity;
      const V4 d4 = mul(transform, V4{0.0f, 0.0f, -1.0f, 0.0f});
      const V3 d = normalize(V3{d4.x, d4.y, d4.z});
      l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z;
      scene_.directional_lights.push_back(l);
    } else if (type == "point") {
      pt_point_light l{};
      std::memcpy(l.color, color, 12);
      l.intensity = intensity;
      l.position[0] = transform.c[3].x; l.position[1] = transform.c[3].y; l.position[2] = transform.c[3].z;
      scene_.point_lights.push_back(l);
    }  // spot lights are ignored, as in the reference
  }

  // Loader::process_camera, gltf_loader.cc:457-485
  void process_camera() {
    const Value* cameras = doc_.find("cameras");
    if (!cameras || cameras->size() == 0) {
      V3 p{-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity()};
      for (const pt_vertex& v : scene_.vertices) {
        p.x = (p.x < v.position[0]) ? v.position[0] : p.x;
        p.y = (p.y < v.position[1]) ? v.position[1] : p.y;
        p.z = (p.z < v.position[2]) ? v.position[2] : p.z;
      }
      p = {p.x * 1.75f, p.y * 1.75f, p.z * 1.75f};
      const V3 look = normalize(V3{-p.x, -p.y, -p.z});
      scene_.camera.position[0] = p.x; scene_.camera.position[1] = p.y; scene_.camera.position[2] = p.z;
      scene_.camera.lookat[0] = look.x; scene_.camera.lookat[1] = look.y; scene_.camera.lookat[2] = look.z;
      scene_.camera.vertical_fov = 45.0f * 0.01745329251994329576923690768489f;  // glm::radians(45.0f)
      return;
    }
    const Value& cam = cameras->array[0];
    const M3 nm = normal_matrix(camera_transform_);
    const V3 dir = normalize(mul(nm, V3{0.0f, 0.0f, -1.0f}));
    scene_.camera.position[0] = camera_transform_.c[3].x;
    scene_.camera.position[1] = camera_transform_.c[3].y;
    scene_.camera.position[2] = camera_transform_.c[3].z;
    scene_.camera.lookat[0] = dir.x; scene_.camera.lookat[1] = dir.y; scene_.camera.lookat[2] = dir.z;
    const Value* persp = cam.find("perspective");
    scene_.camera.vertical_fov = static_cast<float>(persp ? persp->number_or("yfov", 0.7853981633974483) : 0.7853981633974483);
  }

  Scene& scene_;
  std::string base_dir_;
  Value doc_;
  std::vector<unsigned char> glb_bin_;
  std::vector<std::vector<unsigned char>> buffers_;
  uint32_t index_offset_ = 0;
  M4 camera_transform_ = identity();
};

}  // namespace

Scene load_gltf(const std::string& path) {
  Scene scene;
  Loader loader(path, scene);
  loader.process();
  if (scene.triangles.empty()) throw std::runtime_error("glTF: the default scene has no triangles");
  return scene;
}

}  // namespace pt

// ------------------------------------------------------------------ C ABI view of a loaded scene
struct pt_host_scene {
  pt::Scene scene;
  std::vector<pt_image> image_views;
};

static thread_local std::string g_host_error;

extern "C" PT_API pt_host_scene* pt_host_scene_load(const char* gltf_path) {
  try {
    auto* hs = new pt_host_scene();
    hs->scene = pt::load_gltf(gltf_path ? gltf_path : "");
    for (const pt::Image& im : hs->scene.images) hs->image_views.push_back(pt_image{im.width, im.height, im.data.data()});
    return hs;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return nullptr;
  }
}

extern "C" PT_API const char* pt_host_last_error(void) { return g_host_error.c_str(); }

extern "C" PT_API void pt_host_scene_free(pt_host_scene* hs) { delete hs; }

extern "C" PT_API int pt_host_scene_view(const pt_host_scene* hs, pt_host_scene_arrays* out) {
  if (!hs || !out) return PT_ERR_INVALID;
  const pt::Scene& s = hs->scene;
  out->vertices = s.vertices.data();
  out->vertex_count = static_cast<uint32_t>(s.vertices.size());
  out->triangles = reinterpret_cast<const uint32_t*>(s.triangles.data());
  out->triangle_count = static_cast<uint32_t>(s.triangles.size());
  out->materials = s.materials.data();
  out->material_count = static_cast<uint32_t>(s.materials.size());
  out->images = hs->image_views.data();
  out->image_count = static_cast<uint32_t>(hs->image_views.size());
  out->samplers = s.samplers.data();
  out->sampler_count = static_cast<uint32_t>(s.samplers.size());
  out->textures = s.textures.data();
  out->texture_count = static_cast<uint32_t>(s.textures.size());
  out->directional_lights = s.directional_lights.data();
  out->directional_light_array_size = static_cast<uint32_t>(s.directional_lights.size());
  out->point_lights = s.point_lights.data();
  out->point_light_array_size = static_cast<uint32_t>(s.point_lights.size());
  out->camera = s.camera;
  return PT_OK;
}
