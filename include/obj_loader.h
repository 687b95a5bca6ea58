// obj_loader.h — Wavefront OBJ/MTL reader with the API surface the reference uses from its
// vendored tiny_obj_loader.h v2.0.0 (reference include/tiny_obj_loader.h:531-565 ObjReader,
// used at src/main.cpp:51-63 and :1606-1626).  This is an independent implementation (the vendored
// header is not copied); `namespace tinyobj = objio` below lets the reference's glue code
//     tinyobj::ObjReaderConfig cfg; tinyobj::ObjReader reader;
//     reader.ParseFromFile(path, cfg); reader.GetAttrib(); reader.GetShapes();
// compile unchanged.  Output equality with the vendored loader on the shipped resources is pinned
// by tests/golden/ingest_golden.json (tests/test_host.py).
//
// Semantics reproduced (tiny_obj_loader.h defaults: triangulate = true, vertex_color = true,
// real_t = float):
//   v x y z [r g b]   vn x y z   vt u [v [w]]   f v[/vt][/vn] ...   o / g name   usemtl   mtllib
//   1-based indices, negative = relative to the elements read so far
//   shapes split at every `o` / `g`; `usemtl` only changes the per-face material id
//   polygons: triangles kept; quads split along the SHORTER diagonal ((0,1,2),(0,2,3) if
//   |v2-v0|^2 < |v3-v1|^2 else (0,1,3),(1,2,3)) as tiny_obj_loader.h:1428-1520 does;
//   n > 4: ear clipping on the dominant-axis projection (own implementation, not bit-pinned —
//   no shipped resource contains such faces).
#ifndef OBJ_LOADER_H
#define OBJ_LOADER_H

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace objio {

typedef float real_t;

struct index_t {
  int vertex_index = -1;
  int normal_index = -1;
  int texcoord_index = -1;
};

struct mesh_t {
  std::vector<index_t> indices;
  std::vector<unsigned char> num_face_vertices;
  std::vector<int> material_ids;
  std::vector<unsigned int> smoothing_group_ids;
};

struct shape_t {
  std::string name;
  mesh_t mesh;
};

struct attrib_t {
  std::vector<real_t> vertices;   // xyz
  std::vector<real_t> normals;    // xyz
  std::vector<real_t> texcoords;  // uv
  std::vector<real_t> colors;     // rgb (1,1,1 when the file has none)
};

struct material_t {
  std::string name;
  real_t ambient[3] = {0, 0, 0};
  real_t diffuse[3] = {0, 0, 0};
  real_t specular[3] = {0, 0, 0};
  real_t emission[3] = {0, 0, 0};
  real_t shininess = 1.0f;
  real_t ior = 1.0f;
  real_t dissolve = 1.0f;
  int illum = 0;
  std::string diffuse_texname;
};

struct ObjReaderConfig {
  bool triangulate = true;
  std::string triangulation_method = "simple";
  bool vertex_color = true;
  std::string mtl_search_path;
};

namespace detail {

inline const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') p++; return p; }
inline bool at_end(const char* p) { return *p == '\0' || *p == '\n' || *p == '\r' || *p == '#'; }

inline real_t parse_real(const char*& p, real_t def = 0.0f) {
  p = skip_ws(p);
  if (at_end(p)) return def;
  char* end = nullptr;
  double v = strtod(p, &end);
  if (end == p) { while (!at_end(p) && *p != ' ' && *p != '\t') p++; return def; }
  p = end;
  return (real_t)v;
}

inline bool fix_index(int raw, int n, int& out) {
  if (raw > 0) { out = raw - 1; return true; }
  if (raw < 0) { out = n + raw; return true; }
  return false;  // zero is not allowed
}

struct Corner { int v = -1, vt = -1, vn = -1; };

// parses "v", "v/vt", "v//vn", "v/vt/vn"
inline bool parse_corner(const char*& p, int nv, int nvt, int nvn, Corner& c) {
  p = skip_ws(p);
  if (at_end(p)) return false;
  char* end = nullptr;
  long a = strtol(p, &end, 10);
  if (end == p) return false;
  if (!fix_index((int)a, nv, c.v)) return false;
  p = end;
  if (*p != '/') return true;
  p++;
  if (*p != '/') {
    long b = strtol(p, &end, 10);
    if (end != p) { fix_index((int)b, nvt, c.vt); p = end; }
  }
  if (*p == '/') {
    p++;
    long d = strtol(p, &end, 10);
    if (end != p) { fix_index((int)d, nvn, c.vn); p = end; }
  }
  return true;
}

inline std::string rest_of_line(const char* p) {
  p = skip_ws(p);
  std::string s(p);
  while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ' || s.back() == '\t')) s.pop_back();
  return s;
}

inline index_t to_index(const Corner& c) { index_t i; i.vertex_index = c.v; i.normal_index = c.vn; i.texcoord_index = c.vt; return i; }

// ear clipping for n > 4 (projected on the plane of the polygon's dominant normal axis)
inline void triangulate_polygon(const std::vector<Corner>& poly, const std::vector<real_t>& v, std::vector<Corner>& out) {
  size_t n = poly.size();
  double nx = 0, ny = 0, nz = 0;  // Newell normal
  auto P = [&](size_t k, int a) -> double { size_t vi = (size_t)poly[k].v; return (3 * vi + 2 < v.size()) ? v[3 * vi + a] : 0.0; };
  for (size_t k = 0; k < n; k++) {
    size_t j = (k + 1) % n;
    nx += (P(k, 1) - P(j, 1)) * (P(k, 2) + P(j, 2));
    ny += (P(k, 2) - P(j, 2)) * (P(k, 0) + P(j, 0));
    nz += (P(k, 0) - P(j, 0)) * (P(k, 1) + P(j, 1));
  }
  int ax0 = 0, ax1 = 1; double sgn = nz;
  if (std::fabs(nx) >= std::fabs(ny) && std::fabs(nx) >= std::fabs(nz)) { ax0 = 1; ax1 = 2; sgn = nx; }
  else if (std::fabs(ny) >= std::fabs(nz)) { ax0 = 2; ax1 = 0; sgn = ny; }
  std::vector<size_t> idx(n);
  for (size_t k = 0; k < n; k++) idx[k] = k;
  auto area2 = [&](size_t a, size_t b, size_t c) {
    return (P(b, ax0) - P(a, ax0)) * (P(c, ax1) - P(a, ax1)) - (P(b, ax1) - P(a, ax1)) * (P(c, ax0) - P(a, ax0));
  };
  size_t guard = 0;
  while (idx.size() > 3 && guard++ < 4 * n) {
    bool clipped = false;
    for (size_t k = 0; k < idx.size(); k++) {
      size_t a = idx[(k + idx.size() - 1) % idx.size()], b = idx[k], c = idx[(k + 1) % idx.size()];
      double ar = area2(a, b, c) * (sgn >= 0 ? 1.0 : -1.0);
      if (ar <= 0) continue;  // reflex corner
      bool inside = false;
      for (size_t m : idx) {
        if (m == a || m == b || m == c) continue;
        double s = sgn >= 0 ? 1.0 : -1.0;
        if (area2(a, b, m) * s >= 0 && area2(b, c, m) * s >= 0 && area2(c, a, m) * s >= 0) { inside = true; break; }
      }
      if (inside) continue;
      out.push_back(poly[a]); out.push_back(poly[b]); out.push_back(poly[c]);
      idx.erase(idx.begin() + (long)k);
      clipped = true;
      break;
    }
    if (!clipped) break;
  }
  // remainder (triangle, or a degenerate polygon): fan
  for (size_t k = 1; k + 1 < idx.size(); k++) { out.push_back(poly[idx[0]]); out.push_back(poly[idx[k]]); out.push_back(poly[idx[k + 1]]); }
}

}  // namespace detail

inline bool LoadMtl(std::map<std::string, int>* material_map, std::vector<material_t>* materials, std::istream* in, std::string* warn) {
  (void)warn;
  material_t cur; bool have = false;
  std::string line;
  auto three = [](const char* p, real_t* dst) { for (int k = 0; k < 3; k++) dst[k] = detail::parse_real(p); };
  while (std::getline(*in, line)) {
    const char* p = detail::skip_ws(line.c_str());
    if (detail::at_end(p)) continue;
    if (!strncmp(p, "newmtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
      if (have) { (*material_map)[cur.name] = (int)materials->size(); materials->push_back(cur); }
      cur = material_t(); cur.name = detail::rest_of_line(p + 6); have = true;
    } else if (!strncmp(p, "Ka", 2) && (p[2] == ' ' || p[2] == '\t')) three(p + 2, cur.ambient);
    else if (!strncmp(p, "Kd", 2) && (p[2] == ' ' || p[2] == '\t')) three(p + 2, cur.diffuse);
    else if (!strncmp(p, "Ks", 2) && (p[2] == ' ' || p[2] == '\t')) three(p + 2, cur.specular);
    else if (!strncmp(p, "Ke", 2) && (p[2] == ' ' || p[2] == '\t')) three(p + 2, cur.emission);
    else if (!strncmp(p, "Ns", 2) && (p[2] == ' ' || p[2] == '\t')) { const char* q = p + 2; cur.shininess = detail::parse_real(q); }
    else if (!strncmp(p, "Ni", 2) && (p[2] == ' ' || p[2] == '\t')) { const char* q = p + 2; cur.ior = detail::parse_real(q); }
    else if (p[0] == 'd' && (p[1] == ' ' || p[1] == '\t')) { const char* q = p + 1; cur.dissolve = detail::parse_real(q); }
    else if (!strncmp(p, "illum", 5) && (p[5] == ' ' || p[5] == '\t')) cur.illum = atoi(p + 5);
    else if (!strncmp(p, "map_Kd", 6) && (p[6] == ' ' || p[6] == '\t')) cur.diffuse_texname = detail::rest_of_line(p + 6);
  }
  if (have) { (*material_map)[cur.name] = (int)materials->size(); materials->push_back(cur); }
  return true;
}

class ObjReader {
 public:
  bool ParseFromFile(const std::string& filename, const ObjReaderConfig& config = ObjReaderConfig()) {
    attrib_ = attrib_t(); shapes_.clear(); materials_.clear(); warning_.clear(); error_.clear();
    std::ifstream in(filename.c_str());
    if (!in) { error_ = "Cannot open file [" + filename + "]\n"; valid_ = false; return false; }
    std::string base_dir = config.mtl_search_path;
    if (base_dir.empty()) { size_t s = filename.find_last_of("/\\"); if (s != std::string::npos) base_dir = filename.substr(0, s); }
    valid_ = parse(in, config, base_dir);
    return valid_;
  }
  bool Valid() const { return valid_; }
  const attrib_t& GetAttrib() const { return attrib_; }
  const std::vector<shape_t>& GetShapes() const { return shapes_; }
  const std::vector<material_t>& GetMaterials() const { return materials_; }
  const std::string& Warning() const { return warning_; }
  const std::string& Error() const { return error_; }

 private:
  void flush_shape(shape_t& cur, const std::string& next_name) {
    if (!cur.mesh.indices.empty()) shapes_.push_back(cur);
    cur = shape_t();
    cur.name = next_name;
  }

  bool parse(std::istream& in, const ObjReaderConfig& cfg, const std::string& base_dir) {
    using namespace detail;
    std::map<std::string, int> material_map;
    shape_t cur;
    int material = -1;
    unsigned int smoothing = 0;
    bool any_color = false;
    std::vector<real_t> colors;
    std::string line;
    std::vector<Corner> poly, tri;
    while (std::getline(in, line)) {
      const char* p = skip_ws(line.c_str());
      if (at_end(p)) continue;
      if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
        const char* q = p + 1;
        real_t x = parse_real(q), y = parse_real(q), z = parse_real(q);
        attrib_.vertices.push_back(x); attrib_.vertices.push_back(y); attrib_.vertices.push_back(z);
        const char* t = skip_ws(q);
        if (!at_end(t)) { real_t r = parse_real(q, 1.0f), g = parse_real(q, 1.0f), b = parse_real(q, 1.0f); colors.push_back(r); colors.push_back(g); colors.push_back(b); any_color = true; }
        else { colors.push_back(1.0f); colors.push_back(1.0f); colors.push_back(1.0f); }
      } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
        const char* q = p + 2;
        real_t x = parse_real(q), y = parse_real(q), z = parse_real(q);
        attrib_.normals.push_back(x); attrib_.normals.push_back(y); attrib_.normals.push_back(z);
      } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
        const char* q = p + 2;
        real_t u = parse_real(q), v = parse_real(q);
        attrib_.texcoords.push_back(u); attrib_.texcoords.push_back(v);
      } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
        const char* q = p + 1;
        poly.clear();
        Corner c;
        const int nv = (int)(attrib_.vertices.size() / 3), nvt = (int)(attrib_.texcoords.size() / 2), nvn = (int)(attrib_.normals.size() / 3);
        while (true) {
          c = Corner();
          const char* before = q;
          if (!parse_corner(q, nv, nvt, nvn, c)) {
            q = skip_ws(before);
            if (!at_end(q)) { error_ = "Failed to parse `f' line (zero or malformed index)\n"; return false; }
            break;
          }
          poly.push_back(c);
        }
        if (poly.size() < 3) { warning_ += "Degenerated face found\n."; continue; }
        auto emit_tri = [&](const Corner& a, const Corner& b, const Corner& d) {
          cur.mesh.indices.push_back(to_index(a)); cur.mesh.indices.push_back(to_index(b)); cur.mesh.indices.push_back(to_index(d));
          cur.mesh.num_face_vertices.push_back(3); cur.mesh.material_ids.push_back(material); cur.mesh.smoothing_group_ids.push_back(smoothing);
        };
        if (!cfg.triangulate || poly.size() == 3) {
          if (poly.size() == 3) emit_tri(poly[0], poly[1], poly[2]);
          else {
            for (auto& k : poly) cur.mesh.indices.push_back(to_index(k));
            cur.mesh.num_face_vertices.push_back((unsigned char)poly.size()); cur.mesh.material_ids.push_back(material); cur.mesh.smoothing_group_ids.push_back(smoothing);
          }
        } else if (poly.size() == 4) {
          const std::vector<real_t>& v = attrib_.vertices;
          bool ok = true;
          for (auto& k : poly) if (k.v < 0 || 3 * (size_t)k.v + 2 >= v.size()) ok = false;
          if (!ok) { warning_ += "Face with invalid vertex index found.\n"; continue; }
          auto V = [&](int k, int a) { return v[3 * (size_t)poly[k].v + a]; };
          real_t e02x = V(2, 0) - V(0, 0), e02y = V(2, 1) - V(0, 1), e02z = V(2, 2) - V(0, 2);
          real_t e13x = V(3, 0) - V(1, 0), e13y = V(3, 1) - V(1, 1), e13z = V(3, 2) - V(1, 2);
          real_t sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
          real_t sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
          if (sqr02 < sqr13) { emit_tri(poly[0], poly[1], poly[2]); emit_tri(poly[0], poly[2], poly[3]); }
          else { emit_tri(poly[0], poly[1], poly[3]); emit_tri(poly[1], poly[2], poly[3]); }
        } else {
          tri.clear();
          triangulate_polygon(poly, attrib_.vertices, tri);
          for (size_t k = 0; k + 2 < tri.size(); k += 3) emit_tri(tri[k], tri[k + 1], tri[k + 2]);
        }
      } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || at_end(p + 1))) {
        flush_shape(cur, rest_of_line(p + 1));
      } else if (!strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
        std::string name = rest_of_line(p + 6);
        auto it = material_map.find(name);
        if (it != material_map.end()) material = it->second;
        else { material = -1; warning_ += "material [ '" + name + "' ] not found in .mtl\n"; }
      } else if (!strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
        std::string name = rest_of_line(p + 6);
        std::string path = base_dir.empty() ? name : base_dir + "/" + name;
        std::ifstream mf(path.c_str());
        if (mf) LoadMtl(&material_map, &materials_, &mf, &warning_);
        else warning_ += "Material file [ " + name + " ] not found in a path : " + base_dir + "\n";
      } else if (p[0] == 's' && (p[1] == ' ' || p[1] == '\t')) {
        std::string v = rest_of_line(p + 1);
        smoothing = (v == "off" || v.empty()) ? 0u : (unsigned int)strtoul(v.c_str(), nullptr, 10);
      }
    }
    flush_shape(cur, "");
    if (cfg.vertex_color || any_color) attrib_.colors = colors;
    return true;
  }

  bool valid_ = false;
  attrib_t attrib_;
  std::vector<shape_t> shapes_;
  std::vector<material_t> materials_;
  std::string warning_, error_;
};

}  // namespace objio

// drop-in name for code written against the reference's vendored loader
namespace tinyobj = objio;

#endif  // OBJ_LOADER_H
