// config.h — scene and quality selection of the headless host (host/rt_headless.cpp).
//
// The macro NAMES and their meaning are the reference's (its include/config.h:4-27), so a user who edited that
// file finds the same knobs here; every one of them can also be set from the compiler command line
// (-DMAX_BOUNCE_COUNT=3 ...) or overridden at run time by rt_headless options, because the benchmark
// configurations of BASELINE.md differ in resolution, bounce budget and meshes.
#ifndef RT_CONFIG_H
#define RT_CONFIG_H

// ---- scene ------------------------------------------------------------------------------------------
// mesh in the middle of the scene and mesh orbiting it (Wavefront OBJ, triangulated on load)
#ifndef CENTER_MESH_OBJ_PATH
#define CENTER_MESH_OBJ_PATH "resources/teapot.obj"
#endif
#ifndef ORBITING_MESH_OBJ_PATH
#define ORBITING_MESH_OBJ_PATH "resources/armadillo.obj"   // absent from the reference snapshot: see rt_headless
#endif
// directory holding right/left/top/bottom/front/back.jpg
#ifndef SKYBOX_TEXTURE_DIR
#define SKYBOX_TEXTURE_DIR "resources/skybox_texture_sea"
#endif

// ---- materials: 0 = diffuse (Blinn-Phong + shadow ray), 1 = mirror, 2 = refractive (ior 1.52) -------
#ifndef CENTER_MESH_TYPE
#define CENTER_MESH_TYPE 1
#endif
#ifndef ORBITING_MESH_TYPE
#define ORBITING_MESH_TYPE 0
#endif

// ---- quality ----------------------------------------------------------------------------------------
#ifndef MAX_BOUNCE_COUNT
#define MAX_BOUNCE_COUNT 63          // the ray-generation loop traces MAX_BOUNCE_COUNT + 1 closest-hit rays at most
#endif
#ifndef SAMPLES_PER_PIXEL
#define SAMPLES_PER_PIXEL 4
#endif

// ---- camera controls (consumed by interactive front ends; the headless host scripts its camera) -----
static const float CAMERA_MOUSE_SENSITIVITY = 0.0005f;   // radians per pixel of mouse travel
static const float CAMERA_SPEED = 50.0f;                 // world units per unit of timeParam

// TEST_FPS and VALIDATION_LAYERS_ENABLED of the reference select Vulkan presentation / validation behaviour
// and have no effect here: the headless host always reports Mrays/s and every C-ABI call returns a status.

#endif  // RT_CONFIG_H
