#include "image_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace pt {

namespace {

uint32_t be32(const unsigned char* p) {
  return (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) | (static_cast<uint32_t>(p[2]) << 8) | p[3];
}

int paeth(int a, int b, int c) {
  int p = a + b - c;
  int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

// Undo the PNG row filters of one (sub)image in place. `raw` = height * (1 + row_bytes).
void unfilter(unsigned char* raw, uint32_t height, size_t row_bytes, size_t bpp) {
  const unsigned char* prev = nullptr;
  for (uint32_t y = 0; y < height; ++y) {
    unsigned char* row = raw + y * (row_bytes + 1);
    const unsigned char type = row[0];
    unsigned char* cur = row + 1;
    for (size_t i = 0; i < row_bytes; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0;
      const int b = prev ? prev[i] : 0;
      const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
      int add = 0;
      switch (type) {
        case 0: break;
        case 1: add = a; break;
        case 2: add = b; break;
        case 3: add = (a + b) >> 1; break;
        case 4: add = paeth(a, b, c); break;
        default: throw std::runtime_error("PNG: bad filter type");
      }
      cur[i] = static_cast<unsigned char>(cur[i] + add);
    }
    prev = cur;
  }
}

struct PngInfo {
  uint32_t width = 0, height = 0;
  int depth = 0, color = 0, interlace = 0;
  std::vector<unsigned char> palette;  // rgb triples
  std::vector<unsigned char> trns;
  int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
};

uint32_t sample_at(const unsigned char* row, size_t index, int depth) {
  if (depth == 8) return row[index];
  if (depth == 16) return (static_cast<uint32_t>(row[2 * index]) << 8) | row[2 * index + 1];
  const size_t bit = index * static_cast<size_t>(depth);
  const unsigned char byte = row[bit >> 3];
  const int shift = 8 - depth - static_cast<int>(bit & 7);
  return (byte >> shift) & ((1u << depth) - 1u);
}

// One decoded pixel -> RGBA8 (16-bit samples keep their high byte, low-depth grey is scaled).
void store_pixel(const PngInfo& info, const unsigned char* row, uint32_t x, unsigned char* out) {
  const int ch = info.channels();
  uint32_t s[4] = {0, 0, 0, 0};
  for (int c = 0; c < ch; ++c) s[c] = sample_at(row, static_cast<size_t>(x) * ch + c, info.depth);
  auto to8 = [&](uint32_t v) -> unsigned char {
    if (info.depth == 16) return static_cast<unsigned char>(v >> 8);
    if (info.depth == 8) return static_cast<unsigned char>(v);
    return static_cast<unsigned char>(v * 255u / ((1u << info.depth) - 1u));
  };
  switch (info.color) {
    case 0: {
      unsigned char g = to8(s[0]);
      out[0] = out[1] = out[2] = g;
      out[3] = 255;
      if (info.trns.size() >= 2 && s[0] == ((static_cast<uint32_t>(info.trns[0]) << 8) | info.trns[1])) out[3] = 0;
      break;
    }
    case 2:
      out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]); out[3] = 255;
      if (info.trns.size() >= 6 && s[0] == ((static_cast<uint32_t>(info.trns[0]) << 8) | info.trns[1]) &&
          s[1] == ((static_cast<uint32_t>(info.trns[2]) << 8) | info.trns[3]) &&
          s[2] == ((static_cast<uint32_t>(info.trns[4]) << 8) | info.trns[5]))
        out[3] = 0;
      break;
    case 3: {
      if (3 * s[0] + 2 >= info.palette.size()) throw std::runtime_error("PNG: palette index out of range");
      out[0] = info.palette[3 * s[0]]; out[1] = info.palette[3 * s[0] + 1]; out[2] = info.palette[3 * s[0] + 2];
      out[3] = s[0] < info.trns.size() ? info.trns[s[0]] : 255;
      break;
    }
    case 4:
      out[0] = out[1] = out[2] = to8(s[0]);
      out[3] = to8(s[1]);
      break;
    default:
      out[0] = to8(s[0]); out[1] = to8(s[1]); out[2] = to8(s[2]); out[3] = to8(s[3]);
  }
}

}  // namespace

void decode_image_rgba8(const unsigned char* bytes, size_t size, uint32_t& width, uint32_t& height,
                        std::vector<unsigned char>& rgba) {
  static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (size >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) {  // JPEG: jpeg_decode.cpp (baseline sequential only)
    decode_jpeg_rgba8(bytes, size, width, height, rgba);
    return;
  }
  if (size < 8 || std::memcmp(bytes, kSig, 8) != 0) throw std::runtime_error("unsupported image container (PNG expected)");

  PngInfo info;
  std::vector<unsigned char> idat;
  size_t pos = 8;
  bool seen_ihdr = false, seen_iend = false;
  while (pos + 12 <= size && !seen_iend) {
    const uint32_t len = be32(bytes + pos);
    const unsigned char* type = bytes + pos + 4;
    const unsigned char* data = bytes + pos + 8;
    if (pos + 12 + static_cast<size_t>(len) > size) throw std::runtime_error("PNG: truncated chunk");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error("PNG: bad IHDR");
      info.width = be32(data);
      info.height = be32(data + 4);
      info.depth = data[8];
      info.color = data[9];
      info.interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || info.interlace > 1) throw std::runtime_error("PNG: unsupported method");
      seen_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      info.palette.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      info.trns.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      seen_iend = true;
    }
    pos += 12 + static_cast<size_t>(len);
  }
  if (!seen_ihdr || idat.empty()) throw std::runtime_error("PNG: missing IHDR or IDAT");
  if (!info.width || !info.height || info.width > 32768 || info.height > 32768) throw std::runtime_error("PNG: bad size");
  const int d = info.depth, c = info.color;
  const bool depth_ok = (c == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) ||
                        (c == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                        ((c == 2 || c == 4 || c == 6) && (d == 8 || d == 16));
  if (!depth_ok) throw std::runtime_error("PNG: bad colour type / bit depth");

  const size_t bits_pp = static_cast<size_t>(info.channels()) * static_cast<size_t>(d);
  const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  auto row_bytes_of = [&](uint32_t w) { return (static_cast<size_t>(w) * bits_pp + 7) / 8; };

  // pass geometry: one pass for non-interlaced, seven for Adam7
  struct Pass { uint32_t x0, y0, dx, dy; };
  std::vector<Pass> passes;
  if (info.interlace == 0) passes.push_back({0, 0, 1, 1});
  else passes = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  size_t raw_size = 0;
  for (const Pass& p : passes) {
    const uint32_t w = (info.width + p.dx - 1 - p.x0) / p.dx, h = (info.height + p.dy - 1 - p.y0) / p.dy;
    if (info.width > p.x0 && info.height > p.y0 && w && h) raw_size += static_cast<size_t>(h) * (row_bytes_of(w) + 1);
  }
  std::vector<unsigned char> raw(raw_size);
  uLongf got = static_cast<uLongf>(raw_size);
  if (uncompress(raw.data(), &got, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || got != raw_size)
    throw std::runtime_error("PNG: inflate failed");

  width = info.width;
  height = info.height;
  rgba.assign(static_cast<size_t>(width) * height * 4, 0);
  size_t off = 0;
  for (const Pass& p : passes) {
    if (info.width <= p.x0 || info.height <= p.y0) continue;
    const uint32_t w = (info.width + p.dx - 1 - p.x0) / p.dx, h = (info.height + p.dy - 1 - p.y0) / p.dy;
    if (!w || !h) continue;
    const size_t rb = row_bytes_of(w);
    unfilter(raw.data() + off, h, rb, bpp);
    for (uint32_t y = 0; y < h; ++y) {
      const unsigned char* row = raw.data() + off + y * (rb + 1) + 1;
      for (uint32_t x = 0; x < w; ++x) {
        const size_t ox = p.x0 + static_cast<size_t>(x) * p.dx, oy = p.y0 + static_cast<size_t>(y) * p.dy;
        store_pixel(info, row, x, rgba.data() + 4 * (oy * width + ox));
      }
    }
    off += static_cast<size_t>(h) * (rb + 1);
  }
}

bool write_png_rgba8(const std::string& path, uint32_t width, uint32_t height, const unsigned char* rgba,
                     uint32_t stride_bytes) {
  const size_t rb = static_cast<size_t>(width) * 4;
  std::vector<unsigned char> raw(static_cast<size_t>(height) * (rb + 1));
  for (uint32_t y = 0; y < height; ++y) {  // filter 1 (Sub) after the first pixel works well on renders; 0 is always valid
    unsigned char* out = raw.data() + y * (rb + 1);
    const unsigned char* in = rgba + static_cast<size_t>(y) * stride_bytes;
    out[0] = 1;
    for (size_t i = 0; i < rb; ++i) out[1 + i] = static_cast<unsigned char>(in[i] - (i >= 4 ? in[i - 4] : 0));
  }
  uLongf zsize = compressBound(static_cast<uLong>(raw.size()));
  std::vector<unsigned char> z(zsize);
  if (compress2(z.data(), &zsize, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return false;

  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  auto put_chunk = [&](const char* type, const unsigned char* data, uint32_t len) {
    unsigned char hdr[8] = {static_cast<unsigned char>(len >> 24), static_cast<unsigned char>(len >> 16),
                            static_cast<unsigned char>(len >> 8), static_cast<unsigned char>(len),
                            static_cast<unsigned char>(type[0]), static_cast<unsigned char>(type[1]),
                            static_cast<unsigned char>(type[2]), static_cast<unsigned char>(type[3])};
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    unsigned char tail[4] = {static_cast<unsigned char>(crc >> 24), static_cast<unsigned char>(crc >> 16),
                             static_cast<unsigned char>(crc >> 8), static_cast<unsigned char>(crc)};
    return std::fwrite(hdr, 1, 8, f) == 8 && (!len || std::fwrite(data, 1, len, f) == len) && std::fwrite(tail, 1, 4, f) == 4;
  };
  static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  unsigned char ihdr[13] = {static_cast<unsigned char>(width >> 24), static_cast<unsigned char>(width >> 16),
                            static_cast<unsigned char>(width >> 8), static_cast<unsigned char>(width),
                            static_cast<unsigned char>(height >> 24), static_cast<unsigned char>(height >> 16),
                            static_cast<unsigned char>(height >> 8), static_cast<unsigned char>(height), 8, 6, 0, 0, 0};
  bool ok = std::fwrite(kSig, 1, 8, f) == 8 && put_chunk("IHDR", ihdr, 13) &&
            put_chunk("IDAT", z.data(), static_cast<uint32_t>(zsize)) && put_chunk("IEND", nullptr, 0);
  ok = (std::fclose(f) == 0) && ok;
  return ok;
}

std::vector<unsigned char> base64_decode(const std::string& text) {
  std::vector<unsigned char> out;
  out.reserve(text.size() * 3 / 4);
  uint32_t acc = 0;
  int bits = 0;
  for (char ch : text) {
    int v;
    if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
    else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
    else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
    else if (ch == '+' || ch == '-') v = 62;
    else if (ch == '/' || ch == '_') v = 63;
    else if (ch == '=') break;
    else continue;  // whitespace
    acc = (acc << 6) | static_cast<uint32_t>(v);
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back(static_cast<unsigned char>((acc >> bits) & 0xFF));
    }
  }
  return out;
}

}  // namespace pt

#include "pathtracer_amd.h"

extern "C" int pt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8, uint32_t stride_bytes) {
  if (!path || !rgba8 || !width || !height || stride_bytes < width * 4) return PT_ERR_INVALID;
  return pt::write_png_rgba8(path, width, height, rgba8, stride_bytes) ? PT_OK : PT_ERR_INVALID;
}

extern "C" int pt_decode_image(const uint8_t* bytes, size_t size, uint32_t* width, uint32_t* height, uint8_t* rgba8_out,
                               size_t out_capacity) {
  if (!bytes || !width || !height) return PT_ERR_INVALID;
  try {
    std::vector<unsigned char> rgba;
    pt::decode_image_rgba8(bytes, size, *width, *height, rgba);
    if (rgba8_out) {
      if (out_capacity < rgba.size()) return PT_ERR_LIMIT;
      std::memcpy(rgba8_out, rgba.data(), rgba.size());
    }
  } catch (const std::exception&) {
    return PT_ERR_INVALID;
  }
  return PT_OK;
}
