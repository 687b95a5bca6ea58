// jpeg_decode.h — host-side JPEG reader for the skybox faces (baseline + progressive Huffman, 8 bit).
// Stands where the reference calls stbi_load(..., STBI_rgb_alpha) (src/main.cpp:2073-2080).
#ifndef RT_JPEG_DECODE_H
#define RT_JPEG_DECODE_H
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rtjpeg {

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> rgba;  // w*h*4, row 0 = top, alpha 255
};

bool decode_memory(const uint8_t* data, size_t n, Image& out, std::string& err);
bool decode_file(const char* path, Image& out, std::string& err);

}  // namespace rtjpeg
#endif
