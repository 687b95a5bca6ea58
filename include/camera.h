// camera.h — the reference's yaw/pitch fly camera (reference include/camera.h:16-35,
// src/camera.cpp:8-143) with the same class, enum and method names, on the GLM-free shim rt_vec.h.
#ifndef RT_CAMERA_H
#define RT_CAMERA_H

#include "rt_vec.h"

enum CameraMovementDirection {
  RIGHT = 0,
  LEFT,
  UP,
  DOWN,
  FORWARD,
  BACKWARD,
};

class Camera {
 private:
  rtm::vec3 position, front, up, right;
  float pitch, yaw;
  void updateCameraVectors();

 public:
  Camera(rtm::vec3 initialPosition = rtm::vec3(0.0f, 0.0f, 20.0f));
  rtm::vec3 getFrontVector() { return front; }
  rtm::vec3 getUpVector() { return up; }
  rtm::vec3 getRightVector() { return right; }
  rtm::mat4 getViewingMatrix();
  rtm::mat4 getViewingMatrixWithoutTranslation();
  rtm::vec3 getPosition() { return position; }
  void move(CameraMovementDirection dir, float distance);
  void processMouseMovement(float xoffset, float yoffset);
  void look(CameraMovementDirection dir);
};

#endif
