// config.h — compile-time scene / quality selection with the SAME macro names and meanings as the
// reference's include/config.h:4-27.  The headless host (host/rt_headless.cpp) uses them as
// defaults and adds run-time overrides, because the benchmark configurations differ in resolution,
// bounce budget and meshes.
#ifndef RT_CONFIG_H
#define RT_CONFIG_H

#define SKYBOX_TEXTURE_DIR "resources/skybox_texture_sea"

#define CENTER_MESH_OBJ_PATH "resources/teapot.obj"
#define ORBITING_MESH_OBJ_PATH "resources/armadillo.obj"

/* Object types: 0 - diffuse, 1 - mirror, 2 - refractive */
#define CENTER_MESH_TYPE 1
#define ORBITING_MESH_TYPE 0

const float CAMERA_MOUSE_SENSITIVITY = 0.0005f;
const float CAMERA_SPEED = 50.0f;

/* TEST_FPS: the reference prints frames/s; the headless host always reports Mrays/s instead. */
// #define TEST_FPS
/* VALIDATION_LAYERS_ENABLED has no meaning without Vulkan; kept so that -D builds do not break. */
// #define VALIDATION_LAYERS_ENABLED

#define MAX_BOUNCE_COUNT 63
#define SAMPLES_PER_PIXEL 4

#endif
