// camera.h — yaw/pitch fly camera with the interface the reference's host code uses (class Camera,
// enum CameraMovementDirection and the method names of its include/camera.h:16-35), implemented on the GLM-free
// vector shim rt_vec.h.  Behaviour follows the reference's src/camera.cpp:
//   * starts at (0,0,20) with yaw = -pi/2, pitch = 0, i.e. looking down -z            (src/camera.cpp:8-14)
//   * front = (cos yaw * cos pitch, sin pitch, sin yaw * cos pitch);
//     right = normalize(-front.z, 0, front.x); up = right x front                      (src/camera.cpp:16-25)
//   * pitch is clamped to +-1.57 rad                                                   (src/camera.cpp:6, 96-103)
//   * look(dir) snaps the basis to one of six axis-aligned views                       (src/camera.cpp:108-143)
#ifndef RT_CAMERA_H
#define RT_CAMERA_H

#include "rt_vec.h"

// order matters: front ends index key tables with these values
enum CameraMovementDirection { RIGHT = 0, LEFT, UP, DOWN, FORWARD, BACKWARD };

class Camera {
 public:
  explicit Camera(rtm::vec3 initialPosition = rtm::vec3(0.0f, 0.0f, 20.0f));

  // ---- state the uniform block needs every frame (src/main.cpp:2879-2899)
  rtm::vec3 getPosition() { return eye_; }
  rtm::vec3 getFrontVector() { return basis_.front; }
  rtm::vec3 getRightVector() { return basis_.right; }
  rtm::vec3 getUpVector() { return basis_.up; }

  // ---- motion
  void move(CameraMovementDirection dir, float distance);      // translate along right / up / front
  void processMouseMovement(float xoffset, float yoffset);     // yaw += x, pitch += y (clamped), rebuild the basis
  void look(CameraMovementDirection dir);                      // snap to an axis-aligned view

  // ---- matrices (unused by the ray tracer, kept for front ends that rasterise overlays)
  rtm::mat4 getViewingMatrix();
  rtm::mat4 getViewingMatrixWithoutTranslation();

 private:
  struct Basis { rtm::vec3 front, up, right; };
  static Basis basisFromAngles(float yaw, float pitch);
  void updateCameraVectors() { basis_ = basisFromAngles(yaw_, pitch_); }

  rtm::vec3 eye_;
  Basis basis_;
  float pitch_, yaw_;
};

#endif  // RT_CAMERA_H
