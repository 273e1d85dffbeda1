// Image codecs the reference gets from stb (src/stb/stb_imp.cc): PNG decode for glTF
// textures (via tinygltf -> stb_image, forced to 4 x 8-bit channels) and PNG encode for the
// result (stbi_write_png, src/main.cc:78-83). zlib does the (in/de)flate.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pt {

// Decodes a PNG (any colour type, bit depth 1..16, non-interlaced or Adam7) or a JPEG (baseline or progressive) to RGBA8.
// Throws std::runtime_error on malformed input or an unsupported container / JPEG flavour (arithmetic-coded, lossless, 12-bit).
void decode_image_rgba8(const unsigned char* bytes, size_t size, uint32_t& width, uint32_t& height,
                        std::vector<unsigned char>& rgba);

// RGBA8 -> PNG file (8-bit RGBA, one IDAT, filter chosen per row). Returns false on I/O failure.
bool write_png_rgba8(const std::string& path, uint32_t width, uint32_t height, const unsigned char* rgba,
                     uint32_t stride_bytes);

// Sequential and progressive DCT JPEG (Huffman, 8-bit), grey or YCbCr, any sampling factors, restart intervals: jpeg_decode.cpp
void decode_jpeg_rgba8(const unsigned char* bytes, size_t size, uint32_t& width, uint32_t& height, std::vector<unsigned char>& rgba);

std::vector<unsigned char> base64_decode(const std::string& text);

}  // namespace pt
