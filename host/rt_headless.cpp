// rt_headless.cpp — headless counterpart of the reference's main() (src/main.cpp:796-3062) around the
// ray-tracing stage: config.h defaults -> OBJ ingest -> geometry/BLAS/TLAS -> uniforms -> skybox ->
// per-frame { animate, TLAS refit, camera -> uniforms, trace } -> image files + Mrays/s.
// Everything Vulkan/GLFW-specific (window, swapchain, descriptors, SBT, present) has no counterpart:
// the stage is reached through the C ABI (include/rt_api.h) instead of vkCmdTraceRaysKHR.
//
//   rt_headless [--width W] [--height H] [--frames N] [--dt SECONDS] [--bounce B] [--spp S]
//               [--center OBJ] [--orbiting OBJ] [--center-type T] [--orbiting-type T]
//               [--skybox DIR] [--out PREFIX] [--device D] [--frames-in-flight P] [--rgba8 | --bgra8]
//               [--gpus N [--loopback]]   N GPUs of this node in one process: band sharding + RCCL gather (include/rt_multi.h);
//                                         --loopback = N logical devices on GPU D, shards moved by device copies (no RCCL)
//               [--batch K]               with --gpus: K <= 8 consecutive frames per pass of the pipeline (rtm_set_batch), one gather per pass
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "camera.h"
#include "config.h"
#include "jpeg_decode.h"
#include "rt_host.hpp"
#include "rt_multi.h"

static void writePFM(const std::string& path, const std::vector<float>& rgba, int W, int H) {
  std::ofstream f(path, std::ios::binary);
  f << "PF\n" << W << " " << H << "\n-1.0\n";
  std::vector<float> row((size_t)W * 3);
  for (int y = H - 1; y >= 0; y--) {  // PFM stores the bottom row first
    for (int x = 0; x < W; x++) for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = rgba[((size_t)y * W + x) * 4 + c];
    f.write((const char*)row.data(), (std::streamsize)(row.size() * sizeof(float)));
  }
}
// 8-bit view: what the reference's UNORM swapchain-format storage image holds (src/main.cpp:1899): clamp + round
static void writePPM(const std::string& path, const std::vector<float>& rgba, int W, int H) {
  std::ofstream f(path, std::ios::binary);
  f << "P6\n" << W << " " << H << "\n255\n";
  std::vector<unsigned char> row((size_t)W * 3);
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++)
      for (int c = 0; c < 3; c++) {
        float v = rgba[((size_t)y * W + x) * 4 + c];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        row[(size_t)x * 3 + c] = (unsigned char)std::lround(v * 255.0f);
      }
    f.write((const char*)row.data(), (std::streamsize)row.size());
  }
}

int main(int argc, char** argv) {
  int W = 800, H = 600;  // the reference's window size (src/main.cpp:805)
  int frames = 3, device = 0, inFlight = 1, gpus = 0, blocksPerCu = 0;
  bool rgba8 = false, bgra8 = false, loopback = false;
  int batchK = 1;
  float dt = 1.0f / 60.0f;
  std::string center = CENTER_MESH_OBJ_PATH, orbiting = ORBITING_MESH_OBJ_PATH, skyDir = SKYBOX_TEXTURE_DIR, out = "frame";
  rt_uniforms uniformStructure = rthost::defaultUniforms();
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
    if (a == "--width") W = atoi(next());
    else if (a == "--height") H = atoi(next());
    else if (a == "--frames") frames = atoi(next());
    else if (a == "--dt") dt = (float)atof(next());
    else if (a == "--bounce") uniformStructure.max_bounce_count = (uint32_t)atoi(next());
    else if (a == "--spp") uniformStructure.samples_per_pixel = (uint32_t)atoi(next());
    else if (a == "--center") center = next();
    else if (a == "--orbiting") orbiting = next();
    else if (a == "--center-type") uniformStructure.center_object_type = (uint32_t)atoi(next());
    else if (a == "--orbiting-type") uniformStructure.orbiting_object_type = (uint32_t)atoi(next());
    else if (a == "--skybox") skyDir = next();
    else if (a == "--out") out = next();
    else if (a == "--device") device = atoi(next());
    else if (a == "--gpus") gpus = atoi(next());
    else if (a == "--blocks-per-cu") blocksPerCu = atoi(next());   // persistent traversal grid (experiments; 0 = the library's choice)
    else if (a == "--loopback") loopback = true;
    else if (a == "--batch") batchK = std::max(1, std::min(8, atoi(next())));   // --gpus N: frames per pass (rtm_set_batch: K consecutive frames, own instances each, one gather per pass)
    else if (a == "--bgra8") { rgba8 = true; bgra8 = true; }   // ... in the byte order of a B8G8R8A8 surface (surfaceFormatList[0], src/main.cpp:1204): <out>.bgra holds the raw bytes
    else if (a == "--rgba8") rgba8 = true;   // frames come back in the 8-bit surface format the reference presents (src/main.cpp:1899); needs --frames-in-flight > 1
    else if (a == "--frames-in-flight") inFlight = std::max(1, atoi(next()));   // the reference: swapchain image count, src/main.cpp:1203
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  try {
    // More than four frames in flight: give every slot's stream its own hardware queue (HIP maps streams onto 4 by default and
    // reads this once at start-up, before the first HIP call).  A frame that is being copied to the host does not compute, so
    // keeping 4 frames COMPUTING takes 5-6 slots: 1920x1080 animated, every frame copied to host memory, measured per frame:
    // 4 slots / 4 queues 0.98 ms, 6 / 4 0.82, 4 / 8 0.84, 6 / 8 0.73, 8 / 8 0.78.  Never overrides the caller's setting.
    if (inFlight > 4 && gpus == 0) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    // resources/armadillo.obj is absent from the reference snapshot: fall back to the labelled stand-in
    std::string meshLabel = orbiting;
    if (!std::ifstream(orbiting).good() && orbiting.find("armadillo.obj") != std::string::npos) {
      orbiting = "resources/generated/armadillo_standin_f132.obj";
      if (!std::ifstream(orbiting).good()) { if (system("mkdir -p resources/generated") != 0) {} rthost::writeArmadilloStandin(orbiting, 132); }
      meshLabel = "armadillo STAND-IN (geodesic f=132)";
    }
    // OBJ Model, Vertex Buffer, Index Buffer (src/main.cpp:1606-1729)
    rthost::SceneGeometry geometry = rthost::loadScene({center, orbiting});
    uniformStructure.orbiting_object_primitive_offset = geometry.orbitingObjectPrimitiveOffset();  // :1872
    uniformStructure.orbiting_object_vertex_offset = geometry.orbitingObjectVertexOffset();        // :1873

    if (gpus > 0) {
      // ---- several GPUs of one node, one process: include/rt_multi.h (band sharding + one RCCL gather per frame) -------------
      auto check = [](int r, const char* fn, rtm_ctx* m) { if (r) throw std::runtime_error(std::string("RT multi exception: return code ") + std::to_string(r) + " (" + fn + "): " + rtm_last_error(m)); };
      std::vector<int> ids(gpus);
      for (int k = 0; k < gpus; k++) ids[k] = loopback ? device : k;
      rtm_ctx* multi = nullptr;
      check(rtm_create(&multi, gpus, ids.data(), inFlight, loopback ? RTM_LOOPBACK : 0), "rtm_create", nullptr);
      check(rtm_upload_geometry(multi, geometry.vertexBuffer.data(), geometry.vertexBuffer.size(), geometry.indexBuffer.data(), geometry.indexBuffer.size(),
                                geometry.ranges.data(), (int)geometry.ranges.size()), "rtm_upload_geometry", multi);
      for (int m = 0; m < (int)geometry.ranges.size(); m++) check(rtm_build_blas(multi, m), "rtm_build_blas", multi);
      const char* faces[6] = {"right", "left", "top", "bottom", "front", "back"};
      std::vector<std::vector<uint8_t>> sky(6);
      int sw = 0, sh = 0;
      for (int f = 0; f < 6; f++) {
        rtjpeg::Image img; std::string err;
        if (!rtjpeg::decode_file((skyDir + "/" + faces[f] + ".jpg").c_str(), img, err)) throw std::runtime_error("skybox: " + err);
        sky[f].swap(img.rgba); sw = img.w; sh = img.h;
      }
      const uint8_t* fp[6];
      for (int f = 0; f < 6; f++) fp[f] = sky[f].data();
      check(rtm_set_skybox(multi, fp, sw, sh), "rtm_set_skybox", multi);
      if (rgba8) check(rtm_set_param(multi, bgra8 ? "output_bgra8" : "output_rgba8", 1), "rtm_set_param", multi);
      rthost::SceneAnimation animation;
      auto instances = [&]() {
        std::vector<rt_instance> inst(2);
        float t[12];
        for (uint32_t i = 0; i < 2; i++) { rthost::glmToVulkan(animation.glmMatrices[i], t); inst[i] = rthost::createInstance(t, i, i); }
        return inst;
      };
      for (int k = 0; k < inFlight; k++) { auto in = instances(); check(rtm_set_instances(multi, k, in.data(), 2, 0), "rtm_set_instances", multi); }
      std::vector<int> pending(inFlight, 0);     // frames of the pass in flight on that slot
      uint64_t rays = 0; int collected = 0, lastPass = 1;
      const void* px = nullptr;
      float timeParam = 0.f;
      auto collect = [&](int k) {
        rt_stats st{};
        check(rtm_trace_wait(multi, k, &px, &st), "rtm_trace_wait", multi);
        rays += st.rays_primary + st.rays_secondary + st.rays_shadow; collected += pending[k]; lastPass = pending[k]; pending[k] = 0;
      };
      // a pass = batchK consecutive frames (--batch; 1 = frame by frame): every frame animates and refits on its own, the devices render
      // their bands of all of them with the launches of one frame, ONE gather brings them to the root
      const int warm = std::min(frames, inFlight);      // (frames, not passes: the sequence of frames does not depend on --batch)
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      bool timed = false;
      std::vector<int> order;
      for (int frame = 0, pass = 0; frame < frames + warm; pass++) {
        const int k = pass % inFlight;
        if (pending[k]) { collect(k); order.erase(std::find(order.begin(), order.end(), k)); }
        if (!timed && frame >= warm) {
          for (int j : order) collect(j);
          order.clear();
          rays = 0; collected = 0; t0 = std::chrono::steady_clock::now(); timed = true;
        }
        const int b = std::min(batchK, (timed ? frames + warm : warm) - frame);
        if (batchK == 1) {
          timeParam += dt * 0.1f;
          animation.animate(timeParam);
          auto in = instances();
          check(rtm_set_instances(multi, k, in.data(), 2, 1), "rtm_set_instances", multi);   // createTLAS(update = true), src/main.cpp:2853-2861
          check(rtm_set_uniforms(multi, k, &uniformStructure), "rtm_set_uniforms", multi);    // copyData(uniform), src/main.cpp:2901-2903
        } else {
          std::vector<rt_instance> all;
          std::vector<rt_uniforms> us((size_t)b, uniformStructure);
          for (int j = 0; j < b; j++) { timeParam += dt * 0.1f; animation.animate(timeParam); auto in = instances(); all.insert(all.end(), in.begin(), in.end()); }
          check(rtm_set_batch(multi, k, b, all.data(), 2, us.data(), 1), "rtm_set_batch", multi);
        }
        check(rtm_trace_async(multi, k, W, H), "rtm_trace_async", multi);
        pending[k] = b; order.push_back(k);
        frame += b;
      }
      for (int j : order) collect(j);
      px = static_cast<const char*>(px) + (size_t)(lastPass - 1) * (size_t)W * H * (rgba8 ? 4 : 16);   // the last frame of the last pass
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (batchK > 1) printf("passes of %d frames (rtm_set_batch)\n", batchK);
      printf("%d frames on %d %s, %d in flight: %dx%d  %.3f ms per frame  %.1f Mrays/s with every frame gathered on device %d and copied to host memory  mesh: %s\n",
             collected, gpus, loopback ? "logical devices (loopback, one GPU)" : "GPUs (RCCL gather)", inFlight, W, H, ms / collected, rays / (ms * 1e3), ids[0], meshLabel.c_str());
      if (rgba8) {
        const unsigned char* b = static_cast<const unsigned char*>(px);
        std::ofstream f(out + ".ppm", std::ios::binary);
        f << "P6\n" << W << " " << H << "\n255\n";
        std::vector<unsigned char> row((size_t)W * 3);
        for (int y = 0; y < H; y++) {
          for (int x = 0; x < W; x++) for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = b[((size_t)y * W + x) * 4 + (bgra8 ? 2 - c : c)];
          f.write((const char*)row.data(), (std::streamsize)row.size());
        }
      } else {
        const float* pf = static_cast<const float*>(px);
        std::vector<float> image(pf, pf + (size_t)W * H * 4);
        writePFM(out + ".pfm", image, W, H);
        writePPM(out + ".ppm", image, W, H);
      }
      printf("wrote %s.%s\n", out.c_str(), rgba8 ? "ppm" : "pfm / .ppm");
      rtm_destroy(multi);
      return 0;
    }

    rthost::Renderer renderer(device);
    renderer.uploadGeometry(geometry);  // + Bottom Level Acceleration Structures (src/main.cpp:1734-1799)

    // Top Level Acceleration Structure (src/main.cpp:1805-1835)
    rthost::SceneAnimation animation;
    auto makeInstances = [&]() {
      std::vector<rt_instance> inst(2);
      float transformMatrix[12];
      for (uint32_t i = 0; i < 2; i++) {
        rthost::glmToVulkan(animation.glmMatrices[i], transformMatrix);
        inst[i] = rthost::createInstance(transformMatrix, i, i);
      }
      return inst;
    };
    renderer.setInstances(makeInstances(), false);

    // Skybox (src/main.cpp:2064-2080): right, left, top, bottom, front, back
    const char* faces[6] = {"right", "left", "top", "bottom", "front", "back"};
    std::vector<std::vector<uint8_t>> sky(6);
    int sw = 0, sh = 0;
    for (int f = 0; f < 6; f++) {
      rtjpeg::Image img; std::string err;
      if (!rtjpeg::decode_file((skyDir + "/" + faces[f] + ".jpg").c_str(), img, err)) throw std::runtime_error("skybox: " + err);
      sky[f].swap(img.rgba); sw = img.w; sh = img.h;
    }
    renderer.setSkybox(sky, sw, sh);
    renderer.setUniforms(uniformStructure);
    renderer.setTiming(true);

    Camera camera;  // (0,0,20) looking down -z (src/camera.cpp:8-14)
    std::vector<float> image;
    float timeParam = 0.f;
    if (inFlight > 1) {
      // Frames in flight, as the reference's swapchain loop has them (src/main.cpp:2905-2967): one Renderer (frame slot:
      // instances/TLAS, uniforms, queues, pinned output) per frame in flight on ONE shared scene; a frame is submitted, and collected when its Renderer comes
      // round again.  The pixels of every frame land in host memory, so the rate below includes the PCIe copy.
      std::vector<std::unique_ptr<rthost::Renderer>> ring;
      ring.push_back(nullptr);
      for (int k = 1; k < inFlight; k++) {
        ring.emplace_back(new rthost::Renderer(renderer));   // a frame slot on the shared scene: no second copy of geometry, BLAS or cube map
        ring[k]->setInstances(makeInstances(), false);
        ring[k]->setUniforms(uniformStructure);
      }
      auto at = [&](int k) -> rthost::Renderer& { return k == 0 ? renderer : *ring[k]; };
      for (int k = 0; k < inFlight; k++) { at(k).setParam(bgra8 ? "output_bgra8" : "output_rgba8", rgba8 ? 1 : 0); at(k).setTiming(false); if (blocksPerCu > 0) at(k).setParam("trace_blocks_per_cu", blocksPerCu); }
      std::vector<char> pending(inFlight, 0);
      uint64_t rays = 0; int collected = 0;
      const void* px = nullptr;
      double submitMs = 0.0, waitMs = 0.0;
      auto collect = [&](int k) {
        auto w0 = std::chrono::steady_clock::now();
        rt_stats st = at(k).wait(px);
        waitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        rays += st.rays_primary + st.rays_secondary + st.rays_shadow; collected++; pending[k] = 0;
      };
      const int warm = std::min(frames, inFlight);   // the first frame of every context allocates its queues
      std::chrono::steady_clock::time_point t0;
      for (int frame = 0; frame < frames + warm; frame++) {
        const int k = frame % inFlight;
        if (pending[k]) collect(k);
        if (frame == warm) {   // drain, then start the clock
          for (int j = 0; j < inFlight; j++) if (pending[j]) collect(j);
          rays = 0; collected = 0; submitMs = waitMs = 0.0; t0 = std::chrono::steady_clock::now();
        }
        timeParam += dt * 0.1f;
        animation.animate(timeParam);
        at(k).setInstances(makeInstances(), true);   // createTLAS(update = true), src/main.cpp:2853-2861
        at(k).setUniforms(uniformStructure);         // copyData(uniform), src/main.cpp:2901-2903
        { auto s0 = std::chrono::steady_clock::now(); at(k).submit(W, H); submitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s0).count(); }
        pending[k] = 1;
      }
      int last = (frames + warm - 1) % inFlight;
      for (int j = 1; j <= inFlight; j++) { const int k = (last + j) % inFlight; if (pending[k]) collect(k); }   // oldest first; the newest frame is collected last
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("%d frames, %d in flight: %dx%d  %.3f ms per frame  %.1f Mrays/s with every frame copied to host memory as %s  mesh: %s\n", collected, inFlight, W, H,
             ms / collected, rays / (ms * 1e3), rgba8 ? "RGBA8" : "RGBA32F", meshLabel.c_str());
      printf("host time per frame: submit %.3f ms, wait %.3f ms\n", submitMs / collected, waitMs / collected);
      if (rgba8) {
        const unsigned char* b = static_cast<const unsigned char*>(px);
        std::ofstream f(out + ".ppm", std::ios::binary);
        f << "P6\n" << W << " " << H << "\n255\n";
        std::vector<unsigned char> row((size_t)W * 3);
        for (int y = 0; y < H; y++) {
          for (int x = 0; x < W; x++) for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = b[((size_t)y * W + x) * 4 + (bgra8 ? 2 - c : c)];
          f.write((const char*)row.data(), (std::streamsize)row.size());
        }
        if (bgra8) { std::ofstream raw(out + ".bgra", std::ios::binary); raw.write((const char*)b, (std::streamsize)((size_t)W * H * 4)); }
        printf("wrote %s.ppm (8-bit frame as stored by the device%s)\n", out.c_str(), bgra8 ? "; raw B8G8R8A8 bytes in .bgra" : "");
        return 0;
      }
      const float* pf = static_cast<const float*>(px);
      image.assign(pf, pf + (size_t)W * H * 4);
      writePFM(out + ".pfm", image, W, H);
      writePPM(out + ".ppm", image, W, H);
      printf("wrote %s.pfm (float32) and %s.ppm (8-bit clamped view)\n", out.c_str(), out.c_str());
      return 0;
    }
    for (int frame = 0; frame < frames; frame++) {
      // main loop body (src/main.cpp:2795-2949) with a fixed time step instead of the wall clock
      timeParam += dt * 0.1f;
      animation.animate(timeParam);
      renderer.setInstances(makeInstances(), true);  // createTLAS(update = true), src/main.cpp:2853-2861
      renderer.setUniforms(uniformStructure);        // copyData(uniform), src/main.cpp:2901-2903
      auto t0 = std::chrono::steady_clock::now();
      rt_stats st = renderer.trace(W, H, image);
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      uint64_t rays = st.rays_primary + st.rays_secondary + st.rays_shadow;
      printf("frame %d: %dx%d  rays %llu (primary %llu secondary %llu shadow %llu)  gpu %.3f ms  %.1f Mrays/s  (host round trip %.2f ms)  mesh: %s\n", frame, W, H,
             (unsigned long long)rays, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_secondary, (unsigned long long)st.rays_shadow,
             st.ms_frame, rays / (st.ms_frame * 1e3), ms, meshLabel.c_str());
    }
    writePFM(out + ".pfm", image, W, H);
    writePPM(out + ".ppm", image, W, H);
    printf("wrote %s.pfm (float32) and %s.ppm (8-bit clamped view)\n", out.c_str(), out.c_str());
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
