// fly_camera.cpp — implementation of include/camera.h (see that header for the behaviour it reproduces).
#include <cmath>

#include "camera.h"

namespace {

const float kPitchLimit = 1.57f;                 // just short of pi/2: the basis never degenerates
const double kHalfPi = 1.57079632679489661923;

float clampPitch(float p) { return p > kPitchLimit ? kPitchLimit : (p < -kPitchLimit ? -kPitchLimit : p); }

// axis-aligned views of look(): front, up, right per direction, in enum order RIGHT..BACKWARD
struct AxisView { float f[3], u[3], r[3]; };
const AxisView kAxisViews[6] = {
    {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}},     // RIGHT
    {{-1, 0, 0}, {0, 1, 0}, {0, 0, -1}},   // LEFT
    {{0, 1, 0}, {0, 0, 1}, {1, 0, 0}},     // UP
    {{0, -1, 0}, {0, 0, -1}, {1, 0, 0}},   // DOWN
    {{0, 0, -1}, {0, 1, 0}, {1, 0, 0}},    // FORWARD
    {{0, 0, 1}, {0, 1, 0}, {-1, 0, 0}},    // BACKWARD
};
rtm::vec3 v3(const float a[3]) { return rtm::vec3(a[0], a[1], a[2]); }

}  // namespace

Camera::Basis Camera::basisFromAngles(float yaw, float pitch) {
  Basis b;
  const float cp = std::cos(pitch);
  b.front = rtm::vec3(std::cos(yaw) * cp, std::sin(pitch), std::sin(yaw) * cp);
  b.right = rtm::normalize(rtm::vec3(-b.front.z, 0.0f, b.front.x));
  b.up = rtm::cross(b.right, b.front);
  return b;
}

Camera::Camera(rtm::vec3 initialPosition) : eye_(initialPosition), pitch_(0.0f), yaw_((float)-kHalfPi) { updateCameraVectors(); }

void Camera::move(CameraMovementDirection dir, float distance) {
  // RIGHT/LEFT, UP/DOWN and FORWARD/BACKWARD are the +/- pairs of the three basis vectors
  const rtm::vec3 axis = dir <= LEFT ? basis_.right : (dir <= DOWN ? basis_.up : basis_.front);
  const float sign = (dir == RIGHT || dir == UP || dir == FORWARD) ? 1.0f : -1.0f;
  eye_ += (sign * distance) * axis;
}

void Camera::processMouseMovement(float xoffset, float yoffset) {
  yaw_ += xoffset;
  pitch_ = clampPitch(pitch_ + yoffset);
  updateCameraVectors();
}

void Camera::look(CameraMovementDirection dir) {
  const AxisView& v = kAxisViews[(int)dir];
  basis_.front = v3(v.f); basis_.up = v3(v.u); basis_.right = v3(v.r);   // yaw/pitch are left as they were, like the reference
}

rtm::mat4 Camera::getViewingMatrix() { return rtm::lookAt(eye_, eye_ + basis_.front, basis_.up); }
rtm::mat4 Camera::getViewingMatrixWithoutTranslation() { return rtm::lookAt(rtm::vec3(0.0f), basis_.front, basis_.up); }
