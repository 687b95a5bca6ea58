// standin_limbs.cpp — second deterministic stand-in for resources/armadillo.obj (absent from the reference snapshot,
// /root/reference/.MISSING_LARGE_BLOBS), built to be HARD for a BVH where the geodesic blob of standin.cpp is easy:
// a standing figure (torso, head with snout and thin ears, bent arms reaching forward with clawed hands, legs with
// clawed feet, a long curled tail, bumpy shell) given as one implicit surface — a smooth union of capsules and discs —
// and meshed with surface nets.  It is NOT star-shaped: rays cross several surfaces, limbs hide the torso, the gaps
// between arm and body are concave, claws / ears / tail tip are thin, and triangle sizes follow a regular grid as a
// scanner's do.  One closed watertight surface, per-vertex normals from the field gradient, faces `f a//a b//b c//c`.
// See rthost::writeArmadilloLimbs in include/rt_host.hpp.  Every number produced with it says "limbs stand-in".
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

#include "rt_host.hpp"

namespace rthost {
namespace {

struct P3 { double x, y, z; };
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator*(P3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double len(P3 a) { return std::sqrt(dot(a, a)); }

// tapered capsule: radius ra at a, rb at b (linear along the axis)
struct Cap { P3 a, b; double ra, rb; };

inline double sdCap(P3 p, const Cap& c) {
  const P3 pa = p - c.a, ba = c.b - c.a;
  double h = dot(pa, ba) / dot(ba, ba);
  h = h < 0.0 ? 0.0 : (h > 1.0 ? 1.0 : h);
  return len(pa - ba * h) - (c.ra + (c.rb - c.ra) * h);
}
// thin rounded disc (ears, shell rim): centre c, unit normal n, radius R, half thickness t
struct Disc { P3 c, n; double R, t; };
inline double sdDisc(P3 p, const Disc& d) {
  const P3 q = p - d.c;
  const double z = dot(q, d.n);
  const P3 inpl = q - d.n * z;
  const double r = len(inpl) - d.R;
  const double dr = r > 0.0 ? r : 0.0;
  return std::sqrt(dr * dr + z * z) - d.t;
}
inline double smin(double a, double b, double k) {
  const double h = std::max(k - std::fabs(a - b), 0.0) / k;
  return std::min(a, b) - h * h * k * 0.25;
}

struct Figure {
  std::vector<Cap> caps;
  std::vector<Disc> discs;
  Figure() {
    auto cap = [&](P3 a, P3 b, double ra, double rb) { caps.push_back({a, b, ra, rb}); };
    // torso + shell hump on the back
    cap({0, -0.75, 0.0}, {0, 0.85, 0.05}, 1.10, 1.00);
    cap({0, -0.55, -0.45}, {0, 0.75, -0.50}, 0.95, 0.85);
    // neck, head, snout
    cap({0, 1.45, 0.10}, {0, 1.95, 0.30}, 0.48, 0.60);
    cap({0, 1.90, 0.55}, {0, 1.68, 1.30}, 0.34, 0.15);
    for (int s = -1; s <= 1; s += 2) {
      const double x = (double)s;
      // ears: thin discs tilted outwards
      P3 n = {0.35 * x, 0.15, 0.92}; n = n * (1.0 / len(n));
      discs.push_back({{0.52 * x, 2.62, 0.05}, n, 0.36, 0.045});
      // arm: shoulder -> elbow -> wrist, reaching forward (a concave gap between arm and body)
      cap({1.00 * x, 0.95, 0.05}, {1.95 * x, 0.15, 0.30}, 0.40, 0.32);
      cap({1.95 * x, 0.15, 0.30}, {1.70 * x, 0.05, 1.35}, 0.31, 0.24);
      cap({1.70 * x, 0.05, 1.35}, {1.66 * x, 0.03, 1.55}, 0.30, 0.28);   // hand
      for (int f = -1; f <= 1; f++)                                        // three claws
        cap({(1.66 + 0.16 * f) * x, 0.03 + 0.05 * f, 1.65}, {(1.62 + 0.30 * f) * x, -0.12 + 0.10 * f, 2.25}, 0.085, 0.025);
      // leg: hip -> knee -> ankle -> toes
      cap({0.55 * x, -1.05, 0.00}, {0.90 * x, -1.95, 0.50}, 0.55, 0.40);
      cap({0.90 * x, -1.95, 0.50}, {0.82 * x, -2.62, 0.05}, 0.38, 0.27);
      cap({0.82 * x, -2.62, 0.05}, {0.88 * x, -2.74, 0.80}, 0.26, 0.20);
      for (int f = -1; f <= 1; f++)
        cap({(0.88 + 0.15 * f) * x, -2.76, 0.85}, {(0.90 + 0.28 * f) * x, -2.84, 1.35}, 0.075, 0.022);
    }
    // tail: a chain that drops behind the body and curls up to one side
    const P3 t[7] = {{0, -1.10, -0.95}, {0, -1.75, -1.55}, {0.05, -2.35, -2.05}, {0.20, -2.70, -2.60}, {0.50, -2.60, -3.05}, {0.85, -2.20, -3.15}, {1.05, -1.80, -2.95}};
    const double tr[7] = {0.46, 0.38, 0.30, 0.23, 0.16, 0.10, 0.04};
    for (int i = 0; i < 6; i++) cap(t[i], t[i + 1], tr[i], tr[i + 1]);
  }
  // signed field: negative inside.  Lipschitz constant <= 2.5 (smooth unions are 1-Lipschitz, the two bump octaves add at most 0.67 + 0.76).
  double operator()(P3 p) const {
    double d = 1e9;
    for (const Cap& c : caps) d = smin(d, sdCap(p, c), 0.16);
    for (const Disc& e : discs) d = smin(d, sdDisc(p, e), 0.10);
    // plated shell / skin: two octaves of bumps everywhere (a scanned armadillo is rough: plates, scales, muscles)
    const double bumps = std::sin(11.0 * p.x + 0.5) * std::sin(11.0 * p.y + 1.5) * std::sin(11.0 * p.z + 2.5);
    const double scales = std::sin(29.0 * p.x + 2.0) * std::sin(31.0 * p.y + 0.7) * std::sin(27.0 * p.z + 1.1);
    return d - 0.035 * bumps - 0.014 * scales;
  }
};

}  // namespace

size_t writeArmadilloLimbs(const std::string& objPath, int N) {
  if (N < 16 || N > 1024) throw std::runtime_error("writeArmadilloLimbs: resolution must be in 16..1024");
  const Figure fig;
  const double lo = -3.45, hi = 3.45, h = (hi - lo) / N;
  const int M = N + 1;                                    // grid points per axis
  auto gid = [&](int i, int j, int k) { return ((size_t)k * M + j) * M + i; };
  auto pt = [&](int i, int j, int k) { return P3{lo + h * i, lo + h * j, lo + h * k}; };
  std::vector<float> f((size_t)M * M * M, 1.0f);          // far outside by default
  // narrow band: blocks of B^3 points whose centre is farther from the surface than the block can reach are skipped
  const int B = 8, nb = (M + B - 1) / B;
  const double reach = 2.5 * (0.5 * B * h * std::sqrt(3.0)) + 2.0 * h;
  const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto fill = [&](unsigned tid) {
    for (int bk = (int)tid; bk < nb; bk += (int)nthreads)
      for (int bj = 0; bj < nb; bj++)
        for (int bi = 0; bi < nb; bi++) {
          const P3 c = {lo + h * (bi * B + 0.5 * (B - 1)), lo + h * (bj * B + 0.5 * (B - 1)), lo + h * (bk * B + 0.5 * (B - 1))};
          const double dc = fig(c);
          const bool far_block = std::fabs(dc) > reach;
          for (int k = bk * B; k < std::min(M, bk * B + B); k++)
            for (int j = bj * B; j < std::min(M, bj * B + B); j++)
              for (int i = bi * B; i < std::min(M, bi * B + B); i++)
                f[gid(i, j, k)] = far_block ? (dc < 0.0 ? -1.0f : 1.0f) : (float)fig(pt(i, j, k));
        }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(fill, t);
    for (auto& t : th) t.join();
  }
  // surface nets: one vertex per sign-changing cell (mean of its edge crossings), one quad per sign-changing edge
  std::vector<int32_t> cell((size_t)N * N * N, -1);
  auto cid = [&](int i, int j, int k) { return ((size_t)k * N + j) * N + i; };
  std::vector<P3> pos;
  static const int E[12][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  for (int k = 0; k < N; k++)
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        float v[8]; int neg = 0;
        for (int c = 0; c < 8; c++) { v[c] = f[gid(i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2))]; neg += v[c] < 0.0f; }
        if (neg == 0 || neg == 8) continue;
        P3 s = {0, 0, 0}; int n = 0;
        for (const auto& e : E) {
          const float a = v[e[0]], b = v[e[1]];
          if ((a < 0.0f) == (b < 0.0f)) continue;
          const double t = (double)a / ((double)a - (double)b);
          const P3 pa = {(double)(e[0] & 1), (double)((e[0] >> 1) & 1), (double)(e[0] >> 2)}, pb = {(double)(e[1] & 1), (double)((e[1] >> 1) & 1), (double)(e[1] >> 2)};
          s = s + pa + (pb - pa) * t; n++;
        }
        s = s * (1.0 / n);
        cell[cid(i, j, k)] = (int32_t)pos.size();
        pos.push_back({lo + h * (i + s.x), lo + h * (j + s.y), lo + h * (k + s.z)});
      }
  std::vector<uint32_t> tri;   // 3 per triangle
  auto quad = [&](int32_t a, int32_t b, int32_t c, int32_t d, bool flip) {
    if (a < 0 || b < 0 || c < 0 || d < 0) return;   // cannot happen for an interior edge; guards the domain border
    if (flip) std::swap(b, d);
    // split along the shorter diagonal
    const double d0 = len(pos[a] - pos[c]), d1 = len(pos[b] - pos[d]);
    if (d0 <= d1) { tri.insert(tri.end(), {(uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)a, (uint32_t)c, (uint32_t)d}); }
    else { tri.insert(tri.end(), {(uint32_t)a, (uint32_t)b, (uint32_t)d, (uint32_t)b, (uint32_t)c, (uint32_t)d}); }
  };
  for (int k = 1; k < N; k++)
    for (int j = 1; j < N; j++)
      for (int i = 1; i < N; i++) {
        const float v0 = f[gid(i, j, k)];
        const bool in0 = v0 < 0.0f;
        // the three grid edges leaving point (i,j,k) in +x, +y, +z; each is shared by four cells
        if (in0 != (f[gid(i + 1, j, k)] < 0.0f)) quad(cell[cid(i, j - 1, k - 1)], cell[cid(i, j, k - 1)], cell[cid(i, j, k)], cell[cid(i, j - 1, k)], !in0);
        if (in0 != (f[gid(i, j + 1, k)] < 0.0f)) quad(cell[cid(i - 1, j, k - 1)], cell[cid(i - 1, j, k)], cell[cid(i, j, k)], cell[cid(i, j, k - 1)], !in0);
        if (in0 != (f[gid(i, j, k + 1)] < 0.0f)) quad(cell[cid(i - 1, j - 1, k)], cell[cid(i, j - 1, k)], cell[cid(i, j, k)], cell[cid(i - 1, j, k)], !in0);
      }
  // normals: gradient of the field (central differences), pointing out of the figure
  std::vector<P3> nrm(pos.size());
  const double e = 0.25 * h;
  auto grad = [&](unsigned tid) {
    for (size_t v = tid; v < pos.size(); v += nthreads) {
      const P3 p = pos[v];
      P3 g = {fig({p.x + e, p.y, p.z}) - fig({p.x - e, p.y, p.z}), fig({p.x, p.y + e, p.z}) - fig({p.x, p.y - e, p.z}), fig({p.x, p.y, p.z + e}) - fig({p.x, p.y, p.z - e})};
      const double l = len(g);
      nrm[v] = l > 0.0 ? g * (1.0 / l) : P3{0, 1, 0};
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(grad, t);
    for (auto& t : th) t.join();
  }
  FILE* fp = fopen(objPath.c_str(), "w");
  if (!fp) throw std::runtime_error("writeArmadilloLimbs: cannot open " + objPath);
  fprintf(fp, "# armadillo LIMBS STAND-IN: implicit figure meshed by surface nets, resolution %d, %zu vertices, %zu triangles\n", N, pos.size(), tri.size() / 3);
  fprintf(fp, "# generated by rthost::writeArmadilloLimbs because the reference snapshot lacks resources/armadillo.obj\n");
  fprintf(fp, "mtllib armadillo.mtl\no armadillo\n");
  for (auto& p : pos) fprintf(fp, "v %.6f %.6f %.6f\n", p.x, p.y, p.z);
  for (auto& q : nrm) fprintf(fp, "vn %.4f %.4f %.4f\n", q.x, q.y, q.z);
  fprintf(fp, "usemtl armadillo\ns 1\n");
  for (size_t t = 0; t < tri.size(); t += 3)
    fprintf(fp, "f %u//%u %u//%u %u//%u\n", tri[t] + 1, tri[t] + 1, tri[t + 1] + 1, tri[t + 1] + 1, tri[t + 2] + 1, tri[t + 2] + 1);
  fclose(fp);
  return tri.size() / 3;
}

}  // namespace rthost
