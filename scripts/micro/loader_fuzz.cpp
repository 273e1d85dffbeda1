// Robustness of the front end on damaged files, under ASan + UBSan (scripts/fuzz_loaders.sh): PNG / JPEG files through
// decode_image_rgba8 and .glb files through load_gltf, each truncated, byte-smashed, bit-flipped and block-overwritten a few
// thousand times (seeded). A damaged file must load or be refused with an exception; never read or write out of bounds.
// Usage: loader_fuzz <scratch-file.glb> <iterations> file...
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "image_io.h"
#include "scene.h"

static std::vector<unsigned char> mutate(const std::vector<unsigned char>& src, std::mt19937& g, bool keep_length_fields) {
  std::vector<unsigned char> d = src;
  const unsigned kind = g() % 5u;
  const size_t lo = keep_length_fields && d.size() > 32 ? 20 : 0;  // .glb: mostly leave the 12-byte header and the chunk header alone
  if (kind == 0) {
    d.resize(g() % d.size());
  } else if (kind == 1) {
    for (unsigned k = 0; k < 1 + g() % 4u; ++k) d[lo + g() % (d.size() - lo)] = static_cast<unsigned char>(g());
  } else if (kind == 2) {
    for (unsigned k = 0; k < 1 + g() % 3u; ++k) d[lo + g() % (d.size() - lo)] ^= static_cast<unsigned char>(1u << (g() % 8u));
  } else if (kind == 3) {
    const size_t p = lo + g() % (d.size() - lo), n = g() % 64u;
    for (size_t k = p; k < p + n && k < d.size(); ++k) d[k] = static_cast<unsigned char>(g());
  } else {  // a digit becomes another digit: numbers in the JSON chunk change and the JSON stays well-formed
    for (unsigned tries = 0, done = 0; tries < 4000 && done < 1 + g() % 3u; ++tries) {
      const size_t p = g() % d.size();
      if (d[p] >= '0' && d[p] <= '9') {
        d[p] = static_cast<unsigned char>('0' + g() % 10u);
        ++done;
      }
    }
  }
  return d;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string scratch = argv[1];
  const int iterations = std::atoi(argv[2]);
  int loaded = 0, refused = 0;
  for (int a = 3; a < argc; ++a) {
    const std::string name = argv[a];
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 2;
    std::vector<unsigned char> src;
    int ch;
    while ((ch = fgetc(f)) != EOF) src.push_back(static_cast<unsigned char>(ch));
    fclose(f);
    const bool glb = name.size() > 4 && name.substr(name.size() - 4) == ".glb";
    std::mt19937 g(99u + static_cast<unsigned>(a));
    for (int it = 0; it < iterations; ++it) {
      const std::vector<unsigned char> d = mutate(src, g, glb);
      try {
        if (glb) {
          FILE* o = fopen(scratch.c_str(), "wb");
          if (!o) return 2;
          fwrite(d.data(), 1, d.size(), o);
          fclose(o);
          const pt::Scene s = pt::load_gltf(scratch);
          (void)s;
        } else {
          uint32_t w = 0, h = 0;
          std::vector<unsigned char> out;
          pt::decode_image_rgba8(d.data(), d.size(), w, h, out);
        }
        ++loaded;
      } catch (const std::exception&) {
        ++refused;
      }
    }
  }
  printf("loaded %d, refused %d\n", loaded, refused);
  return 0;
}
