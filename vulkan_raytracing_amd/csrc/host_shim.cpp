// host_shim.cpp — flat C entry points over include/rt_host.hpp, camera.h and the JPEG decoder so the
// Python harness (tests, bench.py) drives the SAME C++ host code as host/rt_headless.cpp.
// Built into librt_host.so with g++ (no HIP); pure host-side ingest, nothing here traces rays.
#include <cstring>
#include <string>
#include <vector>

#include "camera.h"
#include "rt_host.hpp"

extern "C" {

void* rth_scene_load(const char* const* paths, int n, char* err, int errlen) {
  try {
    std::vector<std::string> files(paths, paths + n);
    return new rthost::SceneGeometry(rthost::loadScene(files));
  } catch (const std::exception& e) {
    if (err && errlen > 0) { strncpy(err, e.what(), (size_t)errlen - 1); err[errlen - 1] = 0; }
    return nullptr;
  }
}
void rth_scene_free(void* s) { delete (rthost::SceneGeometry*)s; }
uint64_t rth_scene_n_floats(void* s) { return ((rthost::SceneGeometry*)s)->vertexBuffer.size(); }
uint64_t rth_scene_n_idx(void* s) { return ((rthost::SceneGeometry*)s)->indexBuffer.size(); }
int rth_scene_n_meshes(void* s) { return (int)((rthost::SceneGeometry*)s)->ranges.size(); }
const float* rth_scene_verts(void* s) { return ((rthost::SceneGeometry*)s)->vertexBuffer.data(); }
const uint32_t* rth_scene_idx(void* s) { return ((rthost::SceneGeometry*)s)->indexBuffer.data(); }
const rt_mesh_range* rth_scene_ranges(void* s) { return ((rthost::SceneGeometry*)s)->ranges.data(); }
int rth_scene_n_materials(void* s) { return (int)((rthost::SceneGeometry*)s)->materials.size(); }
const rt_material* rth_scene_materials(void* s) { return ((rthost::SceneGeometry*)s)->materials.data(); }
uint64_t rth_scene_n_prim_material(void* s) { return ((rthost::SceneGeometry*)s)->primMaterial.size(); }
const uint32_t* rth_scene_prim_material(void* s) { return ((rthost::SceneGeometry*)s)->primMaterial.data(); }
uint32_t rth_scene_orbit_prim_offset(void* s) { return ((rthost::SceneGeometry*)s)->orbitingObjectPrimitiveOffset(); }
uint32_t rth_scene_orbit_vert_offset(void* s) { return ((rthost::SceneGeometry*)s)->orbitingObjectVertexOffset(); }

int rth_write_armadillo_standin(const char* path, int frequency) {
  try { rthost::writeArmadilloStandin(path, frequency); return 0; } catch (...) { return 1; }
}

long long rth_write_armadillo_limbs(const char* path, int resolution) {
  try { return (long long)rthost::writeArmadilloLimbs(path, resolution); } catch (...) { return -1; }
}

void rth_default_uniforms(rt_uniforms* u) { *u = rthost::defaultUniforms(); }
void rth_make_instance(const float* transform12, uint32_t objIndex, uint64_t mesh, rt_instance* out) { *out = rthost::createInstance(transform12, objIndex, mesh); }

// animation state = two column-major mat4 (32 floats); out = two row-major 3x4 (24 floats)
void rth_anim_init(float* state32) { rthost::SceneAnimation a; memcpy(state32, a.glmMatrices, sizeof(float) * 32); }
void rth_anim_step(float* state32, float timeParam) {
  rthost::SceneAnimation a; memcpy((void*)a.glmMatrices, state32, sizeof(float) * 32);
  a.animate(timeParam);
  memcpy(state32, a.glmMatrices, sizeof(float) * 32);
}
void rth_anim_transforms(const float* state32, float* out24) {
  rthost::SceneAnimation a; memcpy((void*)a.glmMatrices, state32, sizeof(float) * 32);
  rthost::glmToVulkan(a.glmMatrices[0], out24); rthost::glmToVulkan(a.glmMatrices[1], out24 + 12);
}

// camera (include/camera.h)
void* rth_camera_new(float x, float y, float z) { return new Camera(rtm::vec3(x, y, z)); }
void rth_camera_free(void* c) { delete (Camera*)c; }
void rth_camera_move(void* c, int dir, float distance) { ((Camera*)c)->move((CameraMovementDirection)dir, distance); }
void rth_camera_mouse(void* c, float xoff, float yoff) { ((Camera*)c)->processMouseMovement(xoff, yoff); }
void rth_camera_look(void* c, int dir) { ((Camera*)c)->look((CameraMovementDirection)dir); }
// out12 = position, front, up, right
void rth_camera_get(void* c, float* out12) {
  Camera* cam = (Camera*)c;
  rtm::vec3 v[4] = {cam->getPosition(), cam->getFrontVector(), cam->getUpVector(), cam->getRightVector()};
  for (int i = 0; i < 4; i++) { out12[3 * i] = v[i].x; out12[3 * i + 1] = v[i].y; out12[3 * i + 2] = v[i].z; }
}
// src/main.cpp:2879-2899: copy the camera into the uniform block
void rth_camera_to_uniforms(void* c, rt_uniforms* u) {
  Camera* cam = (Camera*)c;
  rtm::vec3 p = cam->getPosition(), f = cam->getFrontVector(), r = cam->getRightVector(), up = cam->getUpVector();
  u->position[0] = p.x; u->position[1] = p.y; u->position[2] = p.z;
  u->forward[0] = f.x; u->forward[1] = f.y; u->forward[2] = f.z;
  u->right[0] = r.x; u->right[1] = r.y; u->right[2] = r.z;
  u->up[0] = up.x; u->up[1] = up.y; u->up[2] = up.z;
}

}  // extern "C"

// ---- skybox faces: rtjpeg (host/jpeg_decode.cpp) ---------------------------------------------------
#include <cstdlib>

#include "jpeg_decode.h"
extern "C" {
int rth_decode_jpeg(const char* path, void** rgba, int* w, int* h, char* err, int errlen) {
  rtjpeg::Image img; std::string e;
  if (!rtjpeg::decode_file(path, img, e)) {
    if (err && errlen > 0) { strncpy(err, e.c_str(), (size_t)errlen - 1); err[errlen - 1] = 0; }
    return 1;
  }
  void* p = malloc(img.rgba.size());
  if (!p) return 2;
  memcpy(p, img.rgba.data(), img.rgba.size());
  *rgba = p; *w = img.w; *h = img.h;
  return 0;
}
void rth_free(void* p) { free(p); }
}
