// Robustness of the JPEG decoder on damaged streams: every file named on the command line is truncated, byte-smashed,
// bit-flipped and block-overwritten a few thousand times (seeded) and decoded under ASan + UBSan (scripts/fuzz_jpeg.sh).
// A damaged stream must decode to something or be refused with an exception; never read or write out of bounds.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <vector>

#include "image_io.h"

int main(int argc, char** argv) {
  int decoded = 0, refused = 0;
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 2;
    std::vector<unsigned char> src;
    int ch;
    while ((ch = fgetc(f)) != EOF) src.push_back(static_cast<unsigned char>(ch));
    fclose(f);
    std::mt19937 g(1234u + static_cast<unsigned>(a));
    for (int it = 0; it < 4000; ++it) {
      std::vector<unsigned char> d = src;
      const unsigned kind = g() % 4u;
      if (kind == 0) {
        d.resize(g() % d.size());
      } else if (kind == 1) {
        for (unsigned k = 0; k < 1 + g() % 4u; ++k) d[g() % d.size()] = static_cast<unsigned char>(g());
      } else if (kind == 2) {
        for (unsigned k = 0; k < 1 + g() % 3u; ++k) d[g() % d.size()] ^= static_cast<unsigned char>(1u << (g() % 8u));
      } else {
        const size_t p = g() % d.size(), n = g() % 64u;
        for (size_t k = p; k < p + n && k < d.size(); ++k) d[k] = static_cast<unsigned char>(g());
      }
      uint32_t w = 0, h = 0;
      std::vector<unsigned char> out;
      try {
        pt::decode_jpeg_rgba8(d.data(), d.size(), w, h, out);
        ++decoded;
      } catch (const std::exception&) {
        ++refused;
      }
    }
  }
  printf("decoded %d, refused %d\n", decoded, refused);
  return 0;
}
