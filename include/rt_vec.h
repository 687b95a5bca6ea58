// rt_vec.h — the handful of GLM types/functions the reference's host code uses (glm::vec3,
// glm::mat4, normalize, cross, translate, rotate, transpose, lookAt), restated so that camera.h
// and the frame loop build without GLM (not installed in this image).  Column-major mat4 with
// m[col][row] indexing and binary32 arithmetic in GLM's operation order.
#ifndef RT_VEC_H
#define RT_VEC_H
#include <cmath>

namespace rtm {

struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  vec3(double x_, double y_, double z_) : x((float)x_), y((float)y_), z((float)z_) {}
  vec3(int x_, int y_, int z_) : x((float)x_), y((float)y_), z((float)z_) {}
  float& operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
  vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }

struct vec4 {
  float x, y, z, w;
  vec4() : x(0), y(0), z(0), w(0) {}
  vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
  float& operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
};
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct mat4 {
  vec4 c[4];  // columns
  mat4() {}
  explicit mat4(float d) { c[0] = vec4(d, 0, 0, 0); c[1] = vec4(0, d, 0, 0); c[2] = vec4(0, 0, d, 0); c[3] = vec4(0, 0, 0, d); }
  vec4& operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
inline mat4 operator*(const mat4& a, const mat4& b) {
  mat4 r;
  for (int j = 0; j < 4; j++) r[j] = a[0] * b[j][0] + a[1] * b[j][1] + a[2] * b[j][2] + a[3] * b[j][3];
  return r;
}
inline mat4 transpose(const mat4& m) {
  mat4 r;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[i][j] = m[j][i];
  return r;
}
// glm::translate(m, v): m * T(v)
inline mat4 translate(const mat4& m, vec3 v) {
  mat4 r = m;
  r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
  return r;
}
// glm::rotate(m, angle, axis): m * R(angle, axis)
inline mat4 rotate(const mat4& m, float angle, vec3 v) {
  const float a = angle, c = std::cos(a), s = std::sin(a);
  vec3 axis = normalize(v);
  vec3 temp = axis * (1.0f - c);
  float R[3][3];
  R[0][0] = c + temp[0] * axis[0]; R[0][1] = temp[0] * axis[1] + s * axis[2]; R[0][2] = temp[0] * axis[2] - s * axis[1];
  R[1][0] = temp[1] * axis[0] - s * axis[2]; R[1][1] = c + temp[1] * axis[1]; R[1][2] = temp[1] * axis[2] + s * axis[0];
  R[2][0] = temp[2] * axis[0] + s * axis[1]; R[2][1] = temp[2] * axis[1] - s * axis[0]; R[2][2] = c + temp[2] * axis[2];
  mat4 r;
  r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2];
  r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2];
  r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2];
  r[3] = m[3];
  return r;
}
// glm::lookAt (right-handed)
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
  vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
  mat4 r(1.0f);
  r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
  r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
  r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
  r[3][0] = -dot(s, eye); r[3][1] = -dot(u, eye); r[3][2] = dot(f, eye);
  return r;
}

}  // namespace rtm

#ifndef RT_NO_GLM_ALIAS
namespace glm = rtm;  // lets reference-style host code (glm::vec3, glm::translate ...) build as is
#endif
#endif  // RT_VEC_H
