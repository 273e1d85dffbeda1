// JPEG decode (ITU T.81 DCT-based, 8-bit, Huffman: sequential SOF0/SOF1 and progressive SOF2) to RGBA8 — what the
// reference gets from stb_image through tinygltf for .jpg textures (gltf_loader.cc:279-291; Sponza-class assets ship
// them). Written from the standard: 1 (grey) or 3 (YCbCr) components, sampling factors 1..4 in any combination, restart
// intervals, 8- and 16-bit quantisation tables, JFIF colour conversion, triangle-filter ("fancy") upsampling for the
// common 2x1 / 2x2 chroma layouts and replication otherwise; progressive frames by spectral selection and successive
// approximation (annex G: DC first / refinement scans, AC first / refinement scans with end-of-band runs), coefficients
// kept per component until the last scan. Arithmetic-coded, lossless, hierarchical and 12-bit streams are refused with a
// message that names the frame type.
// The decoded pixels of a lossy codec are decoder-defined: IDCT rounding and chroma upsampling differ between stb_image,
// libjpeg and this file by a few code values ("parity unpinned" for JPEG texels; PNG texels are exact).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "image_io.h"

namespace pt {

namespace {

const unsigned char kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
  // canonical code tables (T.81 annex C / F.2.2.3): codes of length l lie in [mincode[l], maxcode[l]]
  int mincode[17], maxcode[18], valptr[17];
  unsigned char values[256];
  bool defined = false;
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int dc_pred = 0;
  int blocks_w = 0, blocks_h = 0;  // padded to whole MCUs
  std::vector<unsigned char> plane;  // blocks_w*8 x blocks_h*8 samples
  std::vector<int16_t> coefs;        // progressive frames: 64 coefficients per block (natural order), filled scan by scan
};

struct BitReader {
  const unsigned char* p;
  const unsigned char* end;
  uint32_t acc = 0;
  int count = 0;
  bool hit_marker = false;

  void fill() {
    while (count <= 24) {
      int byte = 0;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;  // stuffed zero
          } else {
            hit_marker = true;  // a marker: feed zeros from here on (T.81 F.2.2.5)
            byte = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= static_cast<uint32_t>(byte) << (24 - count);
      count += 8;
    }
  }
  int bit() {
    if (count == 0) fill();
    const int b = static_cast<int>(acc >> 31);
    acc <<= 1;
    --count;
    return b;
  }
  int bits(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | bit();
    return v;
  }
  void reset() {
    acc = 0;
    count = 0;
    hit_marker = false;
  }
};

int decode_symbol(BitReader& br, const Huffman& h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.values[h.valptr[l] + code - h.mincode[l]];
  }
  throw std::runtime_error("JPEG: bad Huffman code");
}

int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }  // T.81 F.2.2.1

// Separable 8x8 inverse DCT in double precision, rounded once, level shift +128, clamp (T.81 A.3.3).
void idct_block(const int* coef, const uint16_t* q, unsigned char* out, int stride) {
  static double c[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    init = true;
  }
  double deq[64], tmp[64];
  for (int i = 0; i < 64; ++i) deq[i] = static_cast<double>(coef[i]) * q[i];
  for (int v = 0; v < 8; ++v)  // rows: over u
    for (int x = 0; x < 8; ++x) {
      double s = 0.0;
      for (int u = 0; u < 8; ++u) s += c[x][u] * deq[v * 8 + u];
      tmp[v * 8 + x] = s;
    }
  for (int x = 0; x < 8; ++x)  // columns: over v
    for (int y = 0; y < 8; ++y) {
      double s = 0.0;
      for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
      const long r = std::lround(s + 128.0);
      out[y * stride + x] = static_cast<unsigned char>(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

uint16_t be16(const unsigned char* p) { return static_cast<uint16_t>((p[0] << 8) | p[1]); }

// one output row of a component at full resolution (width samples), from its plane
void upsample_row(const Component& c, int hmax, int vmax, uint32_t y, uint32_t width, std::vector<unsigned char>& row) {
  const int pw = c.blocks_w * 8, ph = c.blocks_h * 8;
  const int sx = hmax / c.h, sy = vmax / c.v;
  const bool even = hmax % c.h == 0 && vmax % c.v == 0;
  row.resize(width);
  if (even && sx == 1 && sy == 1) {
    std::memcpy(row.data(), &c.plane[static_cast<size_t>(y) * pw], width);
    return;
  }
  if (even && sx == 2 && (sy == 1 || sy == 2)) {
    // triangle filter: 3/4 nearer + 1/4 farther sample in each subsampled direction (libjpeg's "fancy" upsampling)
    const int cw = static_cast<int>((width + 1) / 2);
    std::vector<int> line(static_cast<size_t>(cw));
    if (sy == 1) {
      for (int x = 0; x < cw; ++x) line[x] = 4 * c.plane[static_cast<size_t>(y) * pw + x];
    } else {
      const int cy = static_cast<int>(y / 2);
      int other = (y & 1) ? cy + 1 : cy - 1;
      const int chh = ph;
      if (other < 0) other = 0;
      if (other >= chh) other = chh - 1;
      for (int x = 0; x < cw; ++x) line[x] = 3 * c.plane[static_cast<size_t>(cy) * pw + x] + c.plane[static_cast<size_t>(other) * pw + x];
    }
    // `line` is scaled by 4; horizontal pass scales by 4 again -> /16 with rounding
    for (uint32_t x = 0; x < width; ++x) {
      const int cx = static_cast<int>(x / 2);
      int ox = (x & 1) ? cx + 1 : cx - 1;
      if (ox < 0) ox = 0;
      if (ox >= cw) ox = cw - 1;
      row[x] = static_cast<unsigned char>((3 * line[cx] + line[ox] + 8) >> 4);
    }
    return;
  }
  for (uint32_t x = 0; x < width; ++x) {  // any other ratio: nearest sample
    const uint32_t cx = static_cast<uint32_t>(static_cast<uint64_t>(x) * c.h / hmax), cy = static_cast<uint32_t>(static_cast<uint64_t>(y) * c.v / vmax);
    row[x] = c.plane[static_cast<size_t>(cy < static_cast<uint32_t>(ph) ? cy : ph - 1) * pw + (cx < static_cast<uint32_t>(pw) ? cx : pw - 1)];
  }
}

unsigned char clamp8(int v) { return static_cast<unsigned char>(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

void decode_jpeg_rgba8(const unsigned char* bytes, size_t size, uint32_t& width, uint32_t& height, std::vector<unsigned char>& rgba) {
  if (size < 4 || bytes[0] != 0xFF || bytes[1] != 0xD8) throw std::runtime_error("JPEG: missing SOI");
  uint16_t qt[4][64] = {};
  bool qt_defined[4] = {false, false, false, false};
  Huffman dc[4], ac[4];
  std::vector<Component> comps;
  int restart_interval = 0;
  int hmax = 1, vmax = 1;
  bool have_frame = false, adobe = false, progressive = false;
  int adobe_transform = -1;
  size_t pos = 2;

  while (pos + 4 <= size) {
    if (bytes[pos] != 0xFF) throw std::runtime_error("JPEG: marker expected");
    while (pos < size && bytes[pos] == 0xFF) ++pos;  // fill bytes
    if (pos >= size) break;
    const int marker = bytes[pos++];
    if (marker == 0xD9) break;                                        // EOI
    if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;  // no payload
    if (pos + 2 > size) throw std::runtime_error("JPEG: truncated segment");
    const size_t len = be16(bytes + pos);
    if (len < 2 || pos + len > size) throw std::runtime_error("JPEG: truncated segment");
    const unsigned char* d = bytes + pos + 2;
    const size_t n = len - 2;

    if (marker == 0xDB) {  // DQT
      size_t i = 0;
      while (i < n) {
        const int pq = d[i] >> 4, tq = d[i] & 15;
        ++i;
        if (tq > 3 || pq > 1 || i + static_cast<size_t>(64 * (pq + 1)) > n) throw std::runtime_error("JPEG: bad DQT");
        for (int k = 0; k < 64; ++k) {
          qt[tq][kZigZag[k]] = pq ? be16(d + i + 2 * k) : d[i + k];
        }
        i += static_cast<size_t>(64 * (pq + 1));
        qt_defined[tq] = true;
      }
    } else if (marker == 0xC4) {  // DHT
      size_t i = 0;
      while (i < n) {
        if (i + 17 > n) throw std::runtime_error("JPEG: bad DHT");
        const int tc = d[i] >> 4, th = d[i] & 15;
        if (tc > 1 || th > 3) throw std::runtime_error("JPEG: bad DHT");
        Huffman& h = tc ? ac[th] : dc[th];
        int total = 0, code = 0;
        for (int l = 1; l <= 16; ++l) {
          const int cnt = d[i + l];
          h.valptr[l] = total;
          h.mincode[l] = code;
          h.maxcode[l] = cnt ? code + cnt - 1 : -1;
          code = (code + cnt) << 1;
          total += cnt;
        }
        if (total > 256 || i + 17 + static_cast<size_t>(total) > n) throw std::runtime_error("JPEG: bad DHT");
        std::memcpy(h.values, d + i + 17, static_cast<size_t>(total));
        h.defined = true;
        i += 17 + static_cast<size_t>(total);
      }
    } else if (marker == 0xDD) {  // DRI
      if (n < 2) throw std::runtime_error("JPEG: bad DRI");
      restart_interval = be16(d);
    } else if (marker == 0xEE && n >= 12 && !std::memcmp(d, "Adobe", 5)) {
      adobe = true;
      adobe_transform = d[11];
    } else if (marker >= 0xC0 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {  // SOFn
      if (marker != 0xC0 && marker != 0xC1 && marker != 0xC2) {
        const char* what = marker == 0xC3 ? "lossless" : (marker >= 0xC9 ? "arithmetic-coded" : "hierarchical");
        throw std::runtime_error(std::string("JPEG: ") + what + " JPEG (SOF" + std::to_string(marker - 0xC0) +
                                 ") is not supported; re-encode as baseline or progressive Huffman JPEG, or PNG");
      }
      if (have_frame) throw std::runtime_error("JPEG: second frame header");
      progressive = marker == 0xC2;
      if (n < 6) throw std::runtime_error("JPEG: bad SOF");
      if (d[0] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
      height = be16(d + 1);
      width = be16(d + 3);
      const int nc = d[5];
      if (!width || !height || (nc != 1 && nc != 3) || n < static_cast<size_t>(6 + 3 * nc))
        throw std::runtime_error("JPEG: unsupported frame (size 0 or component count not 1 / 3)");
      if (static_cast<uint64_t>(width) * height > (1ull << 28))  // 16384 x 16384: a damaged header must not become a 12 GB allocation
        throw std::runtime_error("JPEG: frame larger than 2^28 pixels");
      comps.assign(static_cast<size_t>(nc), Component());
      for (int k = 0; k < nc; ++k) {
        comps[k].id = d[6 + 3 * k];
        comps[k].h = d[7 + 3 * k] >> 4;
        comps[k].v = d[7 + 3 * k] & 15;
        comps[k].tq = d[8 + 3 * k];
        if (comps[k].h < 1 || comps[k].h > 4 || comps[k].v < 1 || comps[k].v > 4 || comps[k].tq > 3) throw std::runtime_error("JPEG: bad component");
        if (comps[k].h > hmax) hmax = comps[k].h;
        if (comps[k].v > vmax) vmax = comps[k].v;
      }
      have_frame = true;
    } else if (marker == 0xDA) {  // SOS: sequential frames have one scan with all components (or one per component)
      if (!have_frame) throw std::runtime_error("JPEG: SOS before SOF");
      if (n < 1) throw std::runtime_error("JPEG: bad SOS");
      const int ns = d[0];
      if (ns < 1 || ns > static_cast<int>(comps.size()) || n < static_cast<size_t>(1 + 2 * ns + 3)) throw std::runtime_error("JPEG: bad SOS");
      // spectral selection [ss, se] and successive approximation (ah: bit position of the previous scan of this band, al: of this one)
      const int ss = d[1 + 2 * ns], se = d[2 + 2 * ns], ah = d[3 + 2 * ns] >> 4, al = d[3 + 2 * ns] & 15;
      if (progressive && (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || (ah != 0 && ah != al + 1)))
        throw std::runtime_error("JPEG: bad progressive scan parameters");
      const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
      std::vector<int> scan;
      for (int k = 0; k < ns; ++k) {
        int idx = -1;
        for (size_t c = 0; c < comps.size(); ++c)
          if (comps[c].id == d[1 + 2 * k]) idx = static_cast<int>(c);
        if (idx < 0) throw std::runtime_error("JPEG: scan names an unknown component");
        comps[idx].td = d[2 + 2 * k] >> 4;
        comps[idx].ta = d[2 + 2 * k] & 15;
        if (comps[idx].td > 3 || comps[idx].ta > 3 || (need_dc && !dc[comps[idx].td].defined) || (need_ac && !ac[comps[idx].ta].defined) ||
            (!progressive && !qt_defined[comps[idx].tq]))
          throw std::runtime_error("JPEG: scan uses an undefined table");
        scan.push_back(idx);
      }
      const uint32_t mcu_w = 8u * hmax, mcu_h = 8u * vmax;
      const uint32_t mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
      for (Component& c : comps)
        if (c.plane.empty()) {
          c.blocks_w = static_cast<int>(mcus_x) * c.h;
          c.blocks_h = static_cast<int>(mcus_y) * c.v;
          c.plane.assign(static_cast<size_t>(c.blocks_w) * 8 * c.blocks_h * 8, 128);
          if (progressive) c.coefs.assign(static_cast<size_t>(c.blocks_w) * c.blocks_h * 64, 0);
        }
      BitReader br{bytes + pos + len, bytes + size};
      int eobrun = 0;  // progressive AC scans: blocks still to come whose band holds nothing new (T.81 G.1.2.2)
      // refinement of a coefficient that is already non-zero: one correction bit, applied away from zero (G.1.2.3)
      auto refine = [&](int16_t& c) {
        if (br.bit() && (c & (1 << al)) == 0) c = static_cast<int16_t>(c >= 0 ? c + (1 << al) : c - (1 << al));
      };
      auto decode_block_progressive = [&](Component& c, int bx, int by) {
        int16_t* coef = &c.coefs[(static_cast<size_t>(by) * c.blocks_w + bx) * 64];
        if (ss == 0) {  // DC scan
          if (ah == 0) {
            const int t = decode_symbol(br, dc[c.td]);
            if (t > 11) throw std::runtime_error("JPEG: bad DC category");
            c.dc_pred += extend(br.bits(t), t);
            coef[0] = static_cast<int16_t>(c.dc_pred * (1 << al));
          } else if (br.bit()) {
            coef[0] = static_cast<int16_t>(coef[0] | (1 << al));
          }
          return;
        }
        if (ah == 0) {  // AC band, first scan
          if (eobrun > 0) {
            --eobrun;
            return;
          }
          for (int k = ss; k <= se;) {
            const int rs = decode_symbol(br, ac[c.ta]);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
              if (r < 15) {  // EOBn: this block and 2^r - 1 + extra more end here
                eobrun = (1 << r) - 1;
                if (r) eobrun += br.bits(r);
                break;
              }
              k += 16;  // ZRL
              continue;
            }
            k += r;
            if (k > se) throw std::runtime_error("JPEG: AC run past the band");
            coef[kZigZag[k]] = static_cast<int16_t>(extend(br.bits(s), s) * (1 << al));
            ++k;
          }
          return;
        }
        // AC band, refinement scan: new coefficients are +-1 << al, placed after `r` still-zero positions; every
        // already non-zero coefficient passed on the way takes one correction bit
        int k = ss;
        if (eobrun == 0) {
          while (k <= se) {
            const int rs = decode_symbol(br, ac[c.ta]);
            int r = rs >> 4;
            const int s = rs & 15;
            int value = 0;
            if (s == 0) {
              if (r < 15) {
                eobrun = 1 << r;
                if (r) eobrun += br.bits(r);
                break;  // the rest of this block is refined below, as the first block of the run
              }
            } else {
              if (s != 1) throw std::runtime_error("JPEG: bad refinement symbol");
              value = br.bit() ? (1 << al) : -(1 << al);
            }
            while (k <= se) {
              int16_t& z = coef[kZigZag[k]];
              if (z != 0) {
                refine(z);
              } else {
                if (r == 0) break;
                --r;
              }
              ++k;
            }
            if (s) {
              if (k > se) throw std::runtime_error("JPEG: AC run past the band");
              coef[kZigZag[k]] = static_cast<int16_t>(value);
            }
            ++k;
          }
        }
        if (eobrun > 0) {
          for (; k <= se; ++k) {
            int16_t& z = coef[kZigZag[k]];
            if (z != 0) refine(z);
          }
          --eobrun;
        }
      };
      auto decode_block = [&](Component& c, int bx, int by) {
        if (progressive) {
          decode_block_progressive(c, bx, by);
          return;
        }
        int coef[64] = {0};
        const int t = decode_symbol(br, dc[c.td]);
        if (t > 11) throw std::runtime_error("JPEG: bad DC category");
        c.dc_pred += extend(br.bits(t), t);
        coef[0] = c.dc_pred;
        for (int k = 1; k < 64;) {
          const int rs = decode_symbol(br, ac[c.ta]);
          const int r = rs >> 4, s = rs & 15;
          if (s == 0) {
            if (r != 15) break;  // EOB
            k += 16;
            continue;
          }
          k += r;
          if (k > 63) throw std::runtime_error("JPEG: AC run past the block");
          coef[kZigZag[k]] = extend(br.bits(s), s);
          ++k;
        }
        if (bx < c.blocks_w && by < c.blocks_h)
          idct_block(coef, qt[c.tq], &c.plane[(static_cast<size_t>(by) * 8) * (static_cast<size_t>(c.blocks_w) * 8) + static_cast<size_t>(bx) * 8], c.blocks_w * 8);
      };
      uint32_t mcu_count = 0;
      int next_rst = 0;
      auto restart = [&]() {
        // byte-align, expect RSTn
        br.reset();
        const unsigned char* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
          if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) throw std::runtime_error("JPEG: restart marker expected");
          ++q;
        }
        if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) throw std::runtime_error("JPEG: restart marker out of sequence");
        br.p = q + 2;
        next_rst = (next_rst + 1) & 7;
        for (Component& c : comps) c.dc_pred = 0;
        eobrun = 0;
      };
      for (int idx : scan) comps[idx].dc_pred = 0;  // every scan starts its predictions anew (T.81 E.2.4)
      if (ns == 1) {  // non-interleaved: the component's own blocks, row by row, only those that cover the image
        Component& c = comps[scan[0]];
        const int bw = static_cast<int>((static_cast<uint64_t>(width) * c.h + 8u * hmax - 1) / (8u * hmax));
        const int bh = static_cast<int>((static_cast<uint64_t>(height) * c.v + 8u * vmax - 1) / (8u * vmax));
        for (int by = 0; by < bh; ++by)
          for (int bx = 0; bx < bw; ++bx) {
            if (restart_interval && mcu_count && mcu_count % static_cast<uint32_t>(restart_interval) == 0) restart();
            decode_block(c, bx, by);
            ++mcu_count;
          }
      } else {
        for (uint32_t my = 0; my < mcus_y; ++my)
          for (uint32_t mx = 0; mx < mcus_x; ++mx) {
            if (restart_interval && mcu_count && mcu_count % static_cast<uint32_t>(restart_interval) == 0) restart();
            for (int idx : scan) {
              Component& c = comps[idx];
              for (int v = 0; v < c.v; ++v)
                for (int h = 0; h < c.h; ++h) decode_block(c, static_cast<int>(mx) * c.h + h, static_cast<int>(my) * c.v + v);
            }
            ++mcu_count;
          }
      }
      // continue after the entropy-coded data: find the next marker that is not RSTn / stuffed
      const unsigned char* q = br.p;
      while (q + 1 < bytes + size && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
      pos = static_cast<size_t>(q - bytes);
      continue;
    }
    pos += len;
  }
  if (!have_frame || comps.empty() || comps[0].plane.empty()) throw std::runtime_error("JPEG: no image data");
  if (progressive)  // all scans are in: dequantise and transform every block once
    for (Component& c : comps) {
      if (c.plane.empty()) throw std::runtime_error("JPEG: a component has no scan");
      if (!qt_defined[c.tq]) throw std::runtime_error("JPEG: frame uses an undefined quantisation table");
      int block[64];
      for (int by = 0; by < c.blocks_h; ++by)
        for (int bx = 0; bx < c.blocks_w; ++bx) {
          const int16_t* coef = &c.coefs[(static_cast<size_t>(by) * c.blocks_w + bx) * 64];
          for (int i = 0; i < 64; ++i) block[i] = coef[i];
          idct_block(block, qt[c.tq], &c.plane[(static_cast<size_t>(by) * 8) * (static_cast<size_t>(c.blocks_w) * 8) + static_cast<size_t>(bx) * 8], c.blocks_w * 8);
        }
      c.coefs = std::vector<int16_t>();
    }

  rgba.assign(static_cast<size_t>(width) * height * 4, 255);
  std::vector<unsigned char> ry, rcb, rcr;
  const bool ycc = comps.size() == 3 && !(adobe && adobe_transform == 0);  // Adobe transform 0: the three components are R, G, B
  for (uint32_t y = 0; y < height; ++y) {
    upsample_row(comps[0], hmax, vmax, y, width, ry);
    unsigned char* out = &rgba[static_cast<size_t>(y) * width * 4];
    if (comps.size() == 1) {
      for (uint32_t x = 0; x < width; ++x) out[4 * x] = out[4 * x + 1] = out[4 * x + 2] = ry[x];
      continue;
    }
    upsample_row(comps[1], hmax, vmax, y, width, rcb);
    upsample_row(comps[2], hmax, vmax, y, width, rcr);
    for (uint32_t x = 0; x < width; ++x) {
      if (!ycc) {
        out[4 * x] = ry[x]; out[4 * x + 1] = rcb[x]; out[4 * x + 2] = rcr[x];
        continue;
      }
      // JFIF: R = Y + 1.402 (Cr-128), G = Y - 0.344136 (Cb-128) - 0.714136 (Cr-128), B = Y + 1.772 (Cb-128); 16-bit fixed point
      const int Y = ry[x], cb = rcb[x] - 128, cr = rcr[x] - 128;
      out[4 * x] = clamp8(Y + ((91881 * cr + 32768) >> 16));
      out[4 * x + 1] = clamp8(Y - ((22554 * cb + 46802 * cr + 32768) >> 16));
      out[4 * x + 2] = clamp8(Y + ((116130 * cb + 32768) >> 16));
    }
  }
}

}  // namespace pt
