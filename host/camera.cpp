// camera.cpp — behaviour of the reference camera (src/camera.cpp:8-143): initial yaw -pi/2, pitch 0;
// front = (cos y cos p, sin p, sin y cos p); right = normalize(-front.z, 0, front.x); up = right x
// front; pitch clamped to +-1.57.
#include "camera.h"

#include <cmath>

namespace {
const float kPitchLimit = 1.57f;
const double kHalfPi = 1.57079632679489661923;
}

Camera::Camera(rtm::vec3 initialPosition) : position(initialPosition), pitch(0.0f), yaw((float)-kHalfPi) { updateCameraVectors(); }

void Camera::updateCameraVectors() {
  const float cp = std::cos(pitch);
  front.x = std::cos(yaw) * cp;
  front.y = std::sin(pitch);
  front.z = std::sin(yaw) * cp;
  right = rtm::normalize(rtm::vec3(-front.z, 0.0f, front.x));
  up = rtm::cross(right, front);
}

rtm::mat4 Camera::getViewingMatrixWithoutTranslation() { return rtm::lookAt(rtm::vec3(0.0f), front, up); }
rtm::mat4 Camera::getViewingMatrix() { return rtm::lookAt(position, position + front, up); }

void Camera::move(CameraMovementDirection dir, float distance) {
  switch (dir) {
    case RIGHT: position += distance * right; break;
    case LEFT: position -= distance * right; break;
    case UP: position += distance * up; break;
    case DOWN: position -= distance * up; break;
    case FORWARD: position += distance * front; break;
    case BACKWARD: position -= distance * front; break;
  }
}

void Camera::processMouseMovement(float xoffset, float yoffset) {
  yaw += xoffset;
  pitch += yoffset;
  if (pitch > kPitchLimit) pitch = kPitchLimit;
  else if (pitch < -kPitchLimit) pitch = -kPitchLimit;
  updateCameraVectors();
}

void Camera::look(CameraMovementDirection dir) {
  switch (dir) {
    case RIGHT: front = rtm::vec3(1, 0, 0); up = rtm::vec3(0, 1, 0); right = rtm::vec3(0, 0, 1); break;
    case LEFT: front = rtm::vec3(-1, 0, 0); up = rtm::vec3(0, 1, 0); right = rtm::vec3(0, 0, -1); break;
    case UP: front = rtm::vec3(0, 1, 0); up = rtm::vec3(0, 0, 1); right = rtm::vec3(1, 0, 0); break;
    case DOWN: front = rtm::vec3(0, -1, 0); up = rtm::vec3(0, 0, -1); right = rtm::vec3(1, 0, 0); break;
    case FORWARD: front = rtm::vec3(0, 0, -1); up = rtm::vec3(0, 1, 0); right = rtm::vec3(1, 0, 0); break;
    case BACKWARD: front = rtm::vec3(0, 0, 1); up = rtm::vec3(0, 1, 0); right = rtm::vec3(-1, 0, 0); break;
  }
}
