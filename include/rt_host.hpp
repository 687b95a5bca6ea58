// rt_host.hpp — C++ host side above the C ABI (include/rt_api.h): the reference's own host steps
// for the ray-tracing path, restated headless.  Each function cites the reference code it mirrors
// (paths relative to the reference root).  Errors surface as std::runtime_error, as the reference's
// throwExceptionVulkanAPI does (src/main.cpp:138-147).
#ifndef RT_HOST_HPP
#define RT_HOST_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "config.h"
#include "obj_loader.h"
#include "rt_api.h"
#include "rt_vec.h"

namespace rthost {

// src/main.cpp:138-147
inline void throwExceptionRtAPI(int result, const char* functionName, rt_ctx* ctx = nullptr) {
  std::string message = "RT API exception: return code " + std::to_string(result) + " (" + functionName + ")";
  const char* detail = rt_last_error(ctx);
  if (detail && *detail) message += ": " + std::string(detail);
  throw std::runtime_error(message);
}

// src/main.cpp:51-63.  The reference prints reader.Error() and exit(1)s; a library cannot exit the
// process, so the failure is thrown instead (same message).
inline void parseFile(const tinyobj::ObjReaderConfig& reader_config, tinyobj::ObjReader& reader, const char* fileName) {
  if (!reader.ParseFromFile(fileName, reader_config)) {
    std::string msg = "TinyObjReader: " + (reader.Error().empty() ? std::string("failed to parse ") + fileName : reader.Error());
    throw std::runtime_error(msg);
  }
  if (!reader.Warning().empty()) fprintf(stderr, "TinyObjReader: %s", reader.Warning().c_str());
}

// Everything src/main.cpp:1606-1729 builds on the host for bindings 2 and 3.
struct SceneGeometry {
  std::vector<float> vertexBuffer;          // objects concatenated, [px py pz nx ny nz] per vertex
  std::vector<uint32_t> indexBuffer;        // objects concatenated, object-local indices
  std::vector<uint32_t> primitiveCount;     // per object
  std::vector<rt_mesh_range> ranges;        // per object offsets
  // SURVEY.md §8(f) n4 — what the reference's loader parses and its renderer drops (src/main.cpp:1644-1650 keeps only
  // vertex_index): the MTL materials of every object and the material of every triangle.  materials[0] is the reference's
  // hard-coded surface (src/shader.rgen:51-55) for faces without `usemtl`; rt_set_materials(materials, primMaterial) turns
  // them on, nothing else reads them.
  std::vector<rt_material> materials;
  std::vector<uint32_t> primMaterial;       // one per triangle of indexBuffer
  // src/main.cpp:1872-1873
  uint32_t orbitingObjectPrimitiveOffset() const { return ranges.size() > 1 ? (uint32_t)(ranges[1].first_index / 3) : 0; }
  uint32_t orbitingObjectVertexOffset() const { return ranges.size() > 1 ? (uint32_t)ranges[1].first_float : 0; }
};

// The vertex buffer layout of src/main.cpp:1673-1682: six floats per OBJ vertex, position then the normal found at the
// SAME index in attrib.normals (the reference ignores normal_index).  Written as a gather per vertex record.  When the file
// has FEWER `vn` than `v` the reference reads past the normal array (undefined behaviour, e.g. cube_scene.obj: 18 vn for
// 44 v); the defined rule used here instead: such a vertex takes the vn of the last face corner that references it.  With
// at least as many normals as vertices the reference's in-bounds read is reproduced as is.
inline void interleaveVertices(const tinyobj::attrib_t& attrib, const std::vector<tinyobj::shape_t>& shapes, std::vector<float>& out) {
  const std::vector<tinyobj::real_t>& pos = attrib.vertices;
  const std::vector<tinyobj::real_t>& nrm = attrib.normals;
  const size_t nVerts = pos.size() / 3;
  const bool normalsCoverVertices = nrm.size() >= pos.size();
  out.assign(6 * nVerts, 0.0f);
  for (size_t v = 0; v < nVerts; v++) {
    float* rec = &out[6 * v];
    std::copy(pos.begin() + 3 * v, pos.begin() + 3 * v + 3, rec);
    if (normalsCoverVertices) std::copy(nrm.begin() + 3 * v, nrm.begin() + 3 * v + 3, rec + 3);
  }
  if (normalsCoverVertices) return;
  for (const tinyobj::shape_t& shape : shapes)
    for (const tinyobj::index_t& corner : shape.mesh.indices) {
      if (corner.normal_index < 0 || corner.vertex_index < 0) continue;
      const size_t v = (size_t)corner.vertex_index, n = (size_t)corner.normal_index;
      if (3 * n + 2 >= nrm.size() || v >= nVerts) continue;
      std::copy(nrm.begin() + 3 * n, nrm.begin() + 3 * n + 3, &out[6 * v + 3]);
    }
}

// The surface src/shader.rgen:51-55 hard-codes, as a material record.
inline rt_material referenceMaterial() {
  rt_material m{};
  m.ka[0] = 0.1f; m.ka[1] = 0.3f; m.ka[2] = 0.1f; m.ns = 100.0f;
  m.kd[0] = 0.2f; m.kd[1] = 1.0f; m.kd[2] = 0.2f; m.ni = 1.52f;
  m.ks[0] = m.ks[1] = m.ks[2] = 0.8f; m.type = RT_MATERIAL_TYPE_OF_INSTANCE;
  return m;
}
// MTL record -> rt_material.  illum 3 (reflection, ray traced) makes the surface a mirror, illum 4/6/7/9 (glass / refraction)
// refractive; every other model leaves the type to the instance, as the reference's CENTER/ORBITING_MESH_TYPE do.
inline rt_material materialFromMtl(const tinyobj::material_t& t) {
  rt_material m{};
  for (int k = 0; k < 3; k++) { m.ka[k] = t.ambient[k]; m.kd[k] = t.diffuse[k]; m.ks[k] = t.specular[k]; }
  m.ns = t.shininess; m.ni = t.ior > 0.0f ? t.ior : 1.0f;
  m.type = t.illum == 3 ? 1u : ((t.illum == 4 || t.illum == 6 || t.illum == 7 || t.illum == 9) ? 2u : RT_MATERIAL_TYPE_OF_INSTANCE);
  return m;
}

// src/main.cpp:1606-1729 for an arbitrary list of OBJ files ({CENTER, ORBITING} in the reference).
inline SceneGeometry loadScene(const std::vector<std::string>& fileNames) {
  SceneGeometry g;
  g.materials.push_back(referenceMaterial());
  tinyobj::ObjReaderConfig reader_config;
  for (const std::string& fileName : fileNames) {
    tinyobj::ObjReader reader;
    parseFile(reader_config, reader, fileName.c_str());
    const tinyobj::attrib_t& attrib = reader.GetAttrib();
    const std::vector<tinyobj::shape_t>& shapes = reader.GetShapes();
    rt_mesh_range r{};
    r.first_float = g.vertexBuffer.size();
    r.first_index = g.indexBuffer.size();
    uint32_t prims = 0;
    const uint32_t firstMaterial = (uint32_t)g.materials.size();
    for (const tinyobj::material_t& mt : reader.GetMaterials()) g.materials.push_back(materialFromMtl(mt));
    for (const tinyobj::shape_t& shape : shapes) {              // src/main.cpp:1643-1650
      prims += (uint32_t)shape.mesh.num_face_vertices.size();
      for (const tinyobj::index_t& index : shape.mesh.indices) g.indexBuffer.push_back((uint32_t)index.vertex_index);
      for (size_t f = 0; f < shape.mesh.num_face_vertices.size(); f++) {
        const int id = f < shape.mesh.material_ids.size() ? shape.mesh.material_ids[f] : -1;
        g.primMaterial.push_back(id >= 0 && (size_t)id < reader.GetMaterials().size() ? firstMaterial + (uint32_t)id : 0u);
      }
    }
    r.prim_count = prims;
    std::vector<float> tmp;
    interleaveVertices(attrib, shapes, tmp);
    g.vertexBuffer.insert(g.vertexBuffer.end(), tmp.begin(), tmp.end());
    g.primitiveCount.push_back(prims);
    g.ranges.push_back(r);
  }
  return g;
}

// src/main.cpp:245-249
inline void glmToVulkan(rtm::mat4 glmMatrix, float vulkanMatrix[12]) {
  glmMatrix = rtm::transpose(glmMatrix);
  memcpy(vulkanMatrix, &glmMatrix, sizeof(float) * 12);
}

// src/main.cpp:538-551
inline rt_instance createInstance(const float transformMatrix[12], uint32_t objIndex, uint64_t mesh) {
  rt_instance inst{};
  memcpy(inst.transform, transformMatrix, sizeof(inst.transform));
  inst.custom_index_and_mask = (objIndex & 0xFFFFFFu) | (0xFFu << 24);
  inst.sbt_offset_and_flags = 0u | (0x01u << 24);  // VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR
  inst.mesh = mesh;
  return inst;
}

// src/main.cpp:1847-1866 initialisers
inline rt_uniforms defaultUniforms() {
  rt_uniforms u{};
  const float p[4] = {0, 0, 20, 1}, r[4] = {1, 0, 0, 1}, up[4] = {0, 1, 0, 1}, f[4] = {0, 0, -1, 1};
  memcpy(u.position, p, sizeof(p)); memcpy(u.right, r, sizeof(r)); memcpy(u.up, up, sizeof(up)); memcpy(u.forward, f, sizeof(f));
  u.light_position[0] = u.light_position[1] = u.light_position[2] = 5.0f;
  u.light_intensity = 1.0f;
  u.max_bounce_count = MAX_BOUNCE_COUNT;
  u.samples_per_pixel = SAMPLES_PER_PIXEL;
  u.center_object_type = CENTER_MESH_TYPE;
  u.orbiting_object_type = ORBITING_MESH_TYPE;
  return u;
}

// src/main.cpp:1805-1808 and the per-frame animation of :2836-2844 with a caller-supplied timeParam
// (the reference derives it from the wall clock, :2798-2799).
struct SceneAnimation {
  rtm::mat4 glmMatrices[2];
  SceneAnimation() { glmMatrices[0] = rtm::mat4(1.0f); glmMatrices[1] = rtm::translate(rtm::mat4(1.0f), rtm::vec3(0.0f, 0.0f, 5.0f)); }
  void animate(float timeParam) {
    const double pi = 3.14159265358979323846;
    glmMatrices[0] = glmMatrices[0] * rtm::rotate(rtm::mat4(1.0f), float(timeParam * pi * 0.0001), rtm::vec3(0.0f, 1.0f, 0.0f));
    glmMatrices[1] = rtm::translate(rtm::rotate(rtm::translate(rtm::mat4(1.0f), rtm::vec3(0.0f, 0.0f, -5.0f)), float(timeParam * pi), rtm::vec3(0.0f, 1.0f, 0.0f)),
                                    rtm::vec3(0.0f, 0.0f, 10.0f));
  }
};

// Deterministic stand-in for resources/armadillo.obj, which the reference snapshot lacks
// (.MISSING_LARGE_BLOBS): a class-I geodesic icosahedron of frequency n (20 n^2 triangles,
// 10 n^2 + 2 vertices; n = 132 -> 348 480 triangles, the Stanford armadillo has 345 944) displaced by
// a fixed sum of lobes, radius about 3, smooth per-vertex normals, faces written `f a//a b//b c//c`.
void writeArmadilloStandin(const std::string& objPath, int frequency = 132);
// Second stand-in, deliberately NOT star-shaped (host/standin_limbs.cpp): a standing figure with limbs, claws, ears and
// a curled tail as ONE implicit surface meshed by surface nets on a resolution^3 grid; returns the triangle count
// (resolution 306 -> 345 168 triangles, 172 565 vertices, the size of the Stanford armadillo).
size_t writeArmadilloLimbs(const std::string& objPath, int resolution = 306);

// RAII wrapper over the C ABI; every failure throws (src/main.cpp:138-147 behaviour).
class Renderer {
 public:
  explicit Renderer(int device = 0) {
    int r = rt_create(&ctx_, device);
    if (r) throwExceptionRtAPI(r, "rt_create", nullptr);
  }
  // one more frame in flight on the same GPU: shares `scene`'s geometry, BLAS and cube map (rt_create_frame_slot)
  explicit Renderer(Renderer& scene) {
    int r = rt_create_frame_slot(scene.ctx_, &ctx_);
    if (r) throwExceptionRtAPI(r, "rt_create_frame_slot", nullptr);
  }
  ~Renderer() { rt_destroy(ctx_); }
  Renderer(const Renderer&) = delete;
  Renderer& operator=(const Renderer&) = delete;
  rt_ctx* handle() { return ctx_; }
  void uploadGeometry(const SceneGeometry& g) {
    check(rt_upload_geometry(ctx_, g.vertexBuffer.data(), g.vertexBuffer.size(), g.indexBuffer.data(), g.indexBuffer.size(), g.ranges.data(), (int)g.ranges.size()), "rt_upload_geometry");
    for (int m = 0; m < (int)g.ranges.size(); m++) check(rt_build_blas(ctx_, m), "rt_build_blas");
  }
  void setInstances(const std::vector<rt_instance>& inst, bool update) { check(rt_set_instances(ctx_, inst.data(), (int)inst.size(), update ? 1 : 0), "rt_set_instances"); }
  void setUniforms(const rt_uniforms& u) { check(rt_set_uniforms(ctx_, &u), "rt_set_uniforms"); }
  // row n4: shade with the MTL materials of the loaded OBJ files / give every instance its own type
  void setMaterials(const SceneGeometry& g) { check(rt_set_materials(ctx_, g.materials.data(), (int)g.materials.size(), g.primMaterial.data(), g.primMaterial.size()), "rt_set_materials"); }
  void clearMaterials() { check(rt_set_materials(ctx_, nullptr, 0, nullptr, 0), "rt_set_materials"); }
  void setInstanceTypes(const std::vector<uint32_t>& types) { check(rt_set_instance_types(ctx_, types.data(), (int)types.size()), "rt_set_instance_types"); }
  void setSkybox(const std::vector<std::vector<uint8_t>>& faces, int w, int h) {
    const uint8_t* p[6];
    for (int f = 0; f < 6; f++) p[f] = faces[f].data();
    check(rt_set_skybox(ctx_, p, w, h), "rt_set_skybox");
  }
  rt_stats trace(int W, int H, std::vector<float>& rgba) {
    rgba.resize((size_t)W * H * 4);
    rt_stats st{};
    check(rt_trace(ctx_, W, H, rgba.data(), &st), "rt_trace");
    return st;
  }
  // frames in flight (one pending frame per Renderer): submit = vkQueueSubmit + fence, wait = vkWaitForFences
  void submit(int W, int H) { check(rt_trace_async(ctx_, W, H), "rt_trace_async"); }
  // pixels: W*H*4 floats, or W*H*4 bytes after setParam("output_rgba8", 1)
  rt_stats wait(const void*& pixels) { rt_stats st{}; check(rt_trace_wait(ctx_, &pixels, &st), "rt_trace_wait"); return st; }
  void setParam(const char* name, int value) { check(rt_set_param(ctx_, name, value), "rt_set_param"); }
  void setTiming(bool on) { check(rt_set_timing(ctx_, on ? 1 : 0), "rt_set_timing"); }

 private:
  void check(int r, const char* fn) { if (r) throwExceptionRtAPI(r, fn, ctx_); }
  rt_ctx* ctx_ = nullptr;
};

}  // namespace rthost
#endif  // RT_HOST_HPP
