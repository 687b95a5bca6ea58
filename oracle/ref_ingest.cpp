// ref_ingest.cpp — reference-side dumper (TEST INFRASTRUCTURE, authoring container only).
//
// Compiled by oracle/Makefile against the reference's OWN vendored headers where they lie
// (-I/root/reference/include: tiny_obj_loader.h v2.0.0, stb_image.h v2.27); nothing from the
// reference is copied into this repository.  It runs the reference's ingest calls
//   tinyobj::ObjReader::ParseFromFile / GetAttrib / GetShapes   (src/main.cpp:51-63, 1606-1626)
//   stbi_load(path, &w, &h, &c, STBI_rgb_alpha)                  (src/main.cpp:2073-2080)
// and dumps their raw outputs so tests/golden/make_ingest_golden.py can hash them into fixtures
// that pin (a) oracle/ingest.py and (b) the product's own OBJ loader / JPEG decoder.
//
//   ref_ingest obj <file.obj> <out_prefix>   -> <prefix>.vertices.f32 .normals.f32 .vidx.u32
//                                               .nidx.i32 .faces.u32 (faces per shape)
//   ref_ingest jpg <file.jpg> <out.rgba>     -> raw RGBA8, prints "w h channels"
#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

template <class T> static void dump(const std::string& path, const std::vector<T>& v) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: ref_ingest obj|jpg <in> <out>\n"); return 2; }
  std::string mode = argv[1];
  if (mode == "obj") {
    tinyobj::ObjReaderConfig cfg;  // defaults, as in src/main.cpp:1606
    tinyobj::ObjReader reader;
    if (!reader.ParseFromFile(argv[2], cfg)) { fprintf(stderr, "%s\n", reader.Error().c_str()); return 1; }
    const tinyobj::attrib_t& a = reader.GetAttrib();
    const std::vector<tinyobj::shape_t>& shapes = reader.GetShapes();
    std::vector<uint32_t> vidx, faces; std::vector<int32_t> nidx;
    for (const auto& s : shapes) {
      faces.push_back((uint32_t)s.mesh.num_face_vertices.size());
      for (const auto& i : s.mesh.indices) { vidx.push_back((uint32_t)i.vertex_index); nidx.push_back(i.normal_index); }
    }
    std::string p = argv[3];
    dump(p + ".vertices.f32", a.vertices); dump(p + ".normals.f32", a.normals);
    dump(p + ".vidx.u32", vidx); dump(p + ".nidx.i32", nidx); dump(p + ".faces.u32", faces);
    printf("%zu %zu %zu %zu\n", a.vertices.size(), a.normals.size(), vidx.size(), shapes.size());
    return 0;
  }
  if (mode == "jpg") {
    int w, h, c;
    unsigned char* px = stbi_load(argv[2], &w, &h, &c, STBI_rgb_alpha);
    if (!px) { fprintf(stderr, "stbi_load failed\n"); return 1; }
    FILE* f = fopen(argv[3], "wb"); fwrite(px, 1, (size_t)w * h * 4, f); fclose(f);
    printf("%d %d %d\n", w, h, c);
    return 0;
  }
  return 2;
}
