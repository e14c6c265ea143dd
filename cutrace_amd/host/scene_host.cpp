// scene_host.cpp — scene-JSON loader, binary-STL reader and the host-side
// arithmetic that feeds the hot path (camera basis, mesh AABB).
//
// Mirrors, for the default schema only, what the reference's generic schema DSL
// does (inc/loader.hpp:35-781, inc/json_helpers.hpp:20-139,
// inc/default_schema.hpp:404-940): same keys, defaults, mandatory rules and the
// same stderr diagnostics.  Written as a plain key table instead of the
// reference's compile-time either<>/template machinery.
//
// MUST be compiled with -ffp-contract=off: look_at and the AABB are float
// arithmetic whose bits the device kernel consumes.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/cutrace_host.h"
#include "json.hpp"

namespace cj = cutrace::json;

struct ctr_host_scene {
  std::vector<ctr_object> objects;
  std::vector<ctr_triangle> triangles;
  std::vector<ctr_light> lights;
  std::vector<ctr_material> materials;
  ctr_camera cam{};
  ctr_scene_desc desc{};
  bool ok = true;

  void refresh() {
    desc.objects = objects.data();     desc.n_objects = objects.size();
    desc.triangles = triangles.data(); desc.n_triangles = triangles.size();
    desc.lights = lights.data();       desc.n_lights = lights.size();
    desc.materials = materials.data(); desc.n_materials = materials.size();
    desc.cam = cam;
  }
};

namespace {

// ---- vector helpers with the reference's operation order (inc/vector.hpp) ----
inline ctr_vec3 sub(ctr_vec3 a, ctr_vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline ctr_vec3 cross(ctr_vec3 a, ctr_vec3 o) {
  return {a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x};
}
inline float norm(ctr_vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline ctr_vec3 normalized(ctr_vec3 a) {
  float f = 1.0f / norm(a);
  return {f * a.x, f * a.y, f * a.z};
}

// ---- json_helpers.hpp:89-118 equivalents; errors are plain strings -------------
struct err_t { bool bad = false; std::string msg; };

const cj::value *find(const cj::object &o, const char *key) {
  auto it = o.find(key);
  return it == o.end() ? nullptr : &it->second;
}
std::string missing(const char *key) { return std::string("Cannot find key '") + key + "' in object."; }
std::string expected(const char *type_name) { return std::string("Expected a value of type ") + type_name + "."; }

// coerce<float>/coerce<size_t> on a present value (json_helpers.hpp:89-93)
bool as_number(const cj::value &v, const char *type_name, double &out, err_t &e) {
  if (!v.is_number()) { if (!e.bad) { e.bad = true; e.msg = expected(type_name); } return false; }
  out = v.num;
  return true;
}
// loader_argument<name, vector, ...>::load_from (loader.hpp:108-160)
bool as_vector(const cj::value &v, ctr_vec3 &out, err_t &e) {
  if (!v.is_array()) { if (!e.bad) { e.bad = true; e.msg = expected("array"); } return false; }
  const cj::array &a = *v.arr;
  if (a.size() != 3) {
    if (!e.bad) { e.bad = true; e.msg = "Expected a 3-value array, got " + std::to_string(a.size()) + " instead."; }
    return false;
  }
  double c[3];
  for (int i = 0; i < 3; i++)
    if (!as_number(a[(size_t)i], "float", c[i], e)) return false;
  out = {(float)c[0], (float)c[1], (float)c[2]};
  return true;
}

// Argument readers.  Every reader is evaluated (the reference evaluates all
// loader_arguments before fmap_all picks the FIRST error, either.hpp:366-378), and
// `e` keeps only the first failure.
void arg_vector(const cj::object &o, const char *key, bool mandatory, ctr_vec3 def, ctr_vec3 &out, err_t &e) {
  const cj::value *v = find(o, key);
  if (!v) {
    if (mandatory) { if (!e.bad) { e.bad = true; e.msg = missing(key); } }
    else out = def;
    return;
  }
  as_vector(*v, out, e);
}
void arg_float(const cj::object &o, const char *key, bool mandatory, float def, float &out, err_t &e) {
  const cj::value *v = find(o, key);
  if (!v) {
    if (mandatory) { if (!e.bad) { e.bad = true; e.msg = missing(key); } }
    else out = def;
    return;
  }
  double d;
  if (as_number(*v, "float", d, e)) out = (float)d;
}
void arg_size(const cj::object &o, const char *key, uint64_t &out, err_t &e) {
  const cj::value *v = find(o, key);
  if (!v) { if (!e.bad) { e.bad = true; e.msg = missing(key); } return; }
  double d;
  if (as_number(*v, "size_t", d, e)) out = (uint64_t)d;  // (size_t) C cast of the double, json_helpers.hpp:92
}
void arg_string(const cj::object &o, const char *key, std::string &out, err_t &e) {
  const cj::value *v = find(o, key);
  if (!v) { if (!e.bad) { e.bad = true; e.msg = missing(key); } return; }
  if (!v->is_string()) { if (!e.bad) { e.bad = true; e.msg = expected("string"); } return; }
  out = v->str;
}

// mesh::bounding_box (default_schema.hpp:554-586): per-triangle std::min/std::max
// of the three corners, merged with fminf/fmaxf (vector.hpp:164-174).
inline float min3(float a, float b, float c) { return std::min(std::min(a, b), c); }
inline float max3(float a, float b, float c) { return std::max(std::max(a, b), c); }

// all_objects_schema::load_from (loader.hpp:327-333) for the four default object kinds
err_t load_object(const cj::object &o, ctr_host_scene &hs) {
  err_t e;
  std::string type;
  arg_string(o, "type", type, e);
  if (e.bad) return e;
  ctr_object obj{};
  if (type == "triangle") {  // default_schema.hpp:487-501
    obj.type = CTR_OBJ_TRIANGLE;
    arg_vector(o, "p1", true, {}, obj.v0, e);
    arg_vector(o, "p2", true, {}, obj.v1, e);
    arg_vector(o, "p3", true, {}, obj.v2, e);
    arg_size(o, "material", obj.mat_idx, e);
  } else if (type == "mesh") {  // default_schema.hpp:596-606, ctor :516-545
    obj.type = CTR_OBJ_MESH;
    std::string file;
    arg_string(o, "file", file, e);
    arg_size(o, "material", obj.mat_idx, e);
    if (!e.bad) {
      obj.tri_begin = hs.triangles.size();
      int64_t n = ctr_stl_read(file.c_str(), nullptr, 0);
      if (n > 0) {
        hs.triangles.resize(obj.tri_begin + (size_t)n);
        ctr_stl_read(file.c_str(), hs.triangles.data() + obj.tri_begin, (uint64_t)n);
        obj.tri_count = (uint64_t)n;
      } else {
        // The reference's Assimp call silently yields an empty mesh on failure
        // (default_schema.hpp:523); keep going the same way, but say so.
        std::cerr << "Warning: mesh file '" << file << "' could not be read as binary STL; mesh has 0 triangles.\n";
        obj.tri_count = 0;
      }
      ctr_mesh_bounds(hs.triangles.data() + obj.tri_begin, obj.tri_count, &obj.v0, &obj.v1);
    }
  } else if (type == "plane") {  // default_schema.hpp:633-645
    obj.type = CTR_OBJ_PLANE;
    arg_vector(o, "point", true, {}, obj.v0, e);
    arg_vector(o, "normal", true, {}, obj.v1, e);
    arg_size(o, "material", obj.mat_idx, e);
  } else if (type == "sphere") {  // default_schema.hpp:672-684
    obj.type = CTR_OBJ_SPHERE;
    arg_vector(o, "center", true, {}, obj.v0, e);
    arg_float(o, "radius", true, 0.f, obj.f0, e);
    arg_size(o, "material", obj.mat_idx, e);
  } else {
    e.bad = true;
    e.msg = "Type '" + type + "' is invalid.";  // loader.hpp:300
  }
  if (!e.bad) hs.objects.push_back(obj);
  return e;
}

err_t load_light(const cj::object &o, ctr_host_scene &hs) {
  err_t e;
  std::string type;
  arg_string(o, "type", type, e);
  if (e.bad) return e;
  ctr_light l{};
  const ctr_vec3 white{1.0f, 1.0f, 1.0f};
  if (type == "sun") {  // default_schema.hpp:719-729
    l.type = CTR_LIGHT_SUN;
    arg_vector(o, "direction", true, {}, l.v, e);
    arg_vector(o, "color", false, white, l.color, e);
  } else if (type == "point") {  // default_schema.hpp:754-764
    l.type = CTR_LIGHT_POINT;
    arg_vector(o, "point", true, {}, l.v, e);
    arg_vector(o, "color", false, white, l.color, e);
  } else {
    e.bad = true;
    e.msg = "Type '" + type + "' is invalid.";
  }
  if (!e.bad) hs.lights.push_back(l);
  return e;
}

err_t load_material(const cj::object &o, ctr_host_scene &hs) {
  err_t e;
  std::string type;
  arg_string(o, "type", type, e);
  if (e.bad) return e;
  ctr_material m{};
  if (type == "solid") {  // default_schema.hpp:805-821
    m.type = CTR_MAT_PHONG;
    arg_vector(o, "color", true, {}, m.color, e);
    arg_float(o, "specular", false, 0.3f, m.specular, e);
    arg_float(o, "reflect", false, 0.0f, m.reflexivity, e);
    arg_float(o, "phong", false, 32.0f, m.phong_exp, e);
    arg_float(o, "transparency", false, 0.0f, m.transparency, e);
  } else {
    e.bad = true;
    e.msg = "Type '" + type + "' is invalid.";
  }
  if (!e.bad) hs.materials.push_back(m);
  return e;
}

// cam_schema::load_from: all eight keys mandatory (MK_MANDATORY, default_schema.hpp:888-897),
// constructor order (e,u,l,n,f,w,h,ambient) :863, then to_gpu → look_at :870-874
err_t load_camera(const cj::object &o, ctr_host_scene &hs) {
  err_t e;
  ctr_vec3 eye{}, up{}, look{};
  float nearp = 0, farp = 0, ambient = 0;
  uint64_t w = 0, h = 0;
  arg_vector(o, "eye", true, {}, eye, e);
  arg_vector(o, "up", true, {}, up, e);
  arg_vector(o, "look", true, {}, look, e);
  arg_float(o, "near_plane", true, 0, nearp, e);
  arg_float(o, "far_plane", true, 0, farp, e);
  arg_size(o, "width", w, e);
  arg_size(o, "height", h, e);
  arg_float(o, "ambient", true, 0, ambient, e);
  if (!e.bad) {
    hs.cam.near_plane = nearp;
    hs.cam.far_plane = farp;
    hs.cam.ambient = ambient;
    hs.cam.w = w;
    hs.cam.h = h;
    ctr_camera_look_at(&hs.cam, eye, up, look);
  }
  return e;
}

template <typename F>
void load_array(const cj::object &root, const char *key, const char *what, ctr_host_scene &hs, F &&load_one) {
  const cj::value *v = find(root, key);
  std::string top_err;
  if (!v) top_err = missing(key);
  else if (!v->is_array()) top_err = expected("array");
  if (!top_err.empty()) {
    std::cerr << "Could not find '" << key << "' array: " << top_err << ".\n";  // loader.hpp:699-702
    hs.ok = false;
    return;
  }
  const cj::array &arr = *v->arr;
  for (size_t i = 0; i < arr.size(); i++) {
    err_t e;
    if (!arr[i].is_object()) { e.bad = true; e.msg = "Value is not a JSON object."; }  // force_object, json_helpers.hpp:133-136
    else e = load_one(*arr[i].obj, hs);
    if (e.bad) {
      std::cerr << "Error while loading " << what << " #" << i << ": " << e.msg << "\n";  // loader.hpp:694-697
      hs.ok = false;
    }
  }
}

// full_schema::load_from (loader.hpp:679-760)
void load_from(const cj::object &root, ctr_host_scene &hs) {
  hs.ok = true;
  // default camera (default_cam defaults, default_schema.hpp:835-842) until "camera" loads
  hs.cam = ctr_camera{};
  hs.cam.near_plane = 0.1f; hs.cam.far_plane = 100.0f; hs.cam.ambient = 0.1f;
  hs.cam.w = 1920; hs.cam.h = 1080;
  ctr_camera_look_at(&hs.cam, {0, 0, 0}, {0, 1, 0}, {0, 0, 1});

  load_array(root, "objects", "object", hs, load_object);
  load_array(root, "lights", "light", hs, load_light);
  load_array(root, "materials", "material", hs, load_material);

  const cj::value *c = find(root, "camera");
  err_t e;
  if (!c) { e.bad = true; e.msg = missing("camera"); }
  else if (!c->is_object()) { e.bad = true; e.msg = expected("object"); }
  else e = load_camera(*c->obj, hs);
  if (e.bad) {
    std::cerr << "Could not find 'camera' object or it's invalid: " << e.msg << ".\n";  // loader.hpp:744-747
    hs.ok = false;
  }
}

int parse_text(const std::string &text, const std::string &label, ctr_host_scene **out) {
  auto *hs = new ctr_host_scene();
  *out = hs;
  cj::value root;
  cj::parser p(text);
  if (!p.parse(root)) {
    // loader.hpp:768-771.  DEVIATION (documented in DESIGN.md): the reference returns an
    // empty scene here WITHOUT clearing last_was_success and then renders an empty default
    // frame; this loader reports failure instead.
    std::cerr << "Error while loading file '" << label << "': " << p.error << "\n";
    hs->ok = false;
    hs->refresh();
    return CTR_E_PARSE;
  }
  if (!root.is_object()) {
    std::cerr << "Error while loading file '" << label << "': Value is not a JSON object.\n";  // loader.hpp:773-777
    hs->ok = false;
    hs->refresh();
    return CTR_E_PARSE;
  }
  load_from(*root.obj, *hs);
  hs->refresh();
  return hs->ok ? CTR_OK : CTR_E_PARSE;
}

}  // namespace

extern "C" {

void ctr_camera_look_at(ctr_camera *cam, ctr_vec3 eye, ctr_vec3 up_hint, ctr_vec3 look) {
  // cam::look_at, default_schema.hpp:370-374
  cam->pos = eye;
  cam->up = up_hint;
  cam->forward = normalized(sub(look, cam->pos));
  cam->right = normalized(cross(cam->forward, cam->up));
  cam->up = normalized(cross(cam->right, cam->forward));
}

void ctr_mesh_bounds(const ctr_triangle *tris, uint64_t n, ctr_vec3 *bb_min, ctr_vec3 *bb_max) {
  ctr_vec3 mn{INFINITY, INFINITY, INFINITY}, mx{-INFINITY, -INFINITY, -INFINITY};  // bound::incorrect, vector.hpp:180-185
  for (uint64_t i = 0; i < n; i++) {
    const ctr_triangle &t = tris[i];
    mn.x = fminf(mn.x, min3(t.p1.x, t.p2.x, t.p3.x));
    mn.y = fminf(mn.y, min3(t.p1.y, t.p2.y, t.p3.y));
    mn.z = fminf(mn.z, min3(t.p1.z, t.p2.z, t.p3.z));
    mx.x = fmaxf(mx.x, max3(t.p1.x, t.p2.x, t.p3.x));
    mx.y = fmaxf(mx.y, max3(t.p1.y, t.p2.y, t.p3.y));
    mx.z = fmaxf(mx.z, max3(t.p1.z, t.p2.z, t.p3.z));
  }
  *bb_min = mn;
  *bb_max = mx;
}

uint64_t ctr_rows_count(const ctr_rows *rows, uint64_t h) {
  if (!rows || rows->row_end <= rows->row_begin) return h;
  uint64_t br = rows->block_rows ? rows->block_rows : (h ? h : 1);
  uint32_t np = rows->n_parts ? rows->n_parts : 1;
  uint32_t part = rows->n_parts ? rows->part : 0;
  uint64_t end = rows->row_end < h ? rows->row_end : h;
  uint64_t n = 0;
  for (uint64_t y = rows->row_begin; y < end; y++)
    if (((y / br) % np) == part) n++;
  return n;
}

int64_t ctr_stl_read(const char *path, ctr_triangle *tris, uint64_t cap) {
  FILE *f = fopen(path, "rb");
  if (!f) return -CTR_E_IO;
  unsigned char head[84];
  if (fread(head, 1, 84, f) != 84) { fclose(f); return -CTR_E_IO; }
  uint32_t n;
  memcpy(&n, head + 80, 4);
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  if (sz != (long)(84 + 50ull * n)) { fclose(f); return -CTR_E_INVALID; }  // not a binary STL
  if (!tris) { fclose(f); return (int64_t)n; }
  fseek(f, 84, SEEK_SET);
  uint64_t m = n < cap ? n : cap;
  std::vector<unsigned char> buf(50 * (size_t)m);
  if (m && fread(buf.data(), 50, (size_t)m, f) != m) { fclose(f); return -CTR_E_IO; }
  fclose(f);
  for (uint64_t i = 0; i < m; i++) {
    // facet: normal(12) v1(12) v2(12) v3(12) attr(2); v1,v2,v3 → p1,p2,p3 (face index order,
    // default_schema.hpp:535-541)
    memcpy(&tris[i], buf.data() + 50 * i + 12, 36);
  }
  return (int64_t)m;
}

int ctr_stl_write(const char *path, const ctr_triangle *tris, uint64_t n) {
  FILE *f = fopen(path, "wb");
  if (!f) return CTR_E_IO;
  unsigned char head[84] = {0};
  snprintf((char *)head, 80, "cutrace_amd generated mesh");
  uint32_t n32 = (uint32_t)n;
  memcpy(head + 80, &n32, 4);
  fwrite(head, 1, 84, f);
  for (uint64_t i = 0; i < n; i++) {
    unsigned char rec[50] = {0};
    memcpy(rec + 12, &tris[i], 36);
    fwrite(rec, 1, 50, f);
  }
  fclose(f);
  return CTR_OK;
}

int ctr_host_scene_load(const char *json_path, ctr_host_scene **out) {
  if (!json_path || !out) return CTR_E_INVALID;
  std::ifstream strm(json_path, std::ios::binary);
  if (!strm) {
    *out = nullptr;
    std::cerr << "Error while loading file '" << json_path << "': cannot open file\n";
    return CTR_E_IO;
  }
  std::stringstream ss;
  ss << strm.rdbuf();
  return parse_text(ss.str(), json_path, out);
}

int ctr_host_scene_parse(const char *json_text, ctr_host_scene **out) {
  if (!json_text || !out) return CTR_E_INVALID;
  return parse_text(json_text, "<memory>", out);
}

void ctr_host_scene_free(ctr_host_scene *hs) { delete hs; }

const ctr_scene_desc *ctr_host_scene_desc(ctr_host_scene *hs) {
  if (!hs) return nullptr;
  hs->refresh();
  return &hs->desc;
}

void ctr_host_scene_set_size(ctr_host_scene *hs, uint64_t w, uint64_t h) {
  if (!hs) return;
  hs->cam.w = w;
  hs->cam.h = h;
  hs->refresh();
}

int ctr_host_scene_set_material(ctr_host_scene *hs, uint64_t idx, const ctr_material *m) {
  if (!hs || !m || idx >= hs->materials.size()) return CTR_E_INVALID;
  hs->materials[idx] = *m;
  hs->refresh();
  return CTR_OK;
}

void ctr_dump_scene(const ctr_scene_desc *d) {
  // dump_scene_kernel, kernel.hpp:150-166 — same text, produced on the host
  printf(" -> Have %-4llu objects:\n", (unsigned long long)d->n_objects);
  for (uint64_t i = 0; i < d->n_objects; i++)
    printf("  -> Object   #%-4llu has type #%-2llu\n", (unsigned long long)i, (unsigned long long)d->objects[i].type);
  printf(" -> Have %-4llu lights:\n", (unsigned long long)d->n_lights);
  for (uint64_t i = 0; i < d->n_lights; i++)
    printf("  -> Light    #%-4llu has type #%-2llu\n", (unsigned long long)i, (unsigned long long)d->lights[i].type);
  printf(" -> Have %-4llu materials:\n", (unsigned long long)d->n_materials);
  for (uint64_t i = 0; i < d->n_materials; i++)
    printf("  -> Material #%-4llu has type #%-2llu\n", (unsigned long long)i, (unsigned long long)d->materials[i].type);
  fflush(stdout);
}

void ctr_dump_schema(void) {
  // Static text standing in for the RTTI-driven dump_schema() (schema_view.hpp:17-229,
  // called at main.cu:16-19).  Content = the schema the CODE enforces
  // (default_schema.hpp:463-898), not the stale schema.md.
  fputs(
      "Scene schema:\n"
      "  objects:   array of\n"
      "    { type: \"triangle\", p1: vec3, p2: vec3, p3: vec3, material: size_t }\n"
      "    { type: \"mesh\", file: string (binary STL, relative to the CWD), material: size_t }\n"
      "    { type: \"plane\", point: vec3, normal: vec3, material: size_t }\n"
      "    { type: \"sphere\", center: vec3, radius: float, material: size_t }\n"
      "  lights:    array of\n"
      "    { type: \"sun\", direction: vec3, color: vec3 = [1,1,1] }\n"
      "    { type: \"point\", point: vec3, color: vec3 = [1,1,1] }\n"
      "  materials: array of\n"
      "    { type: \"solid\", color: vec3, specular: float = 0.3, reflect: float = 0,\n"
      "      phong: float = 32, transparency: float = 0 }\n"
      "  camera:    { eye: vec3, up: vec3, look: vec3, near_plane: float, far_plane: float,\n"
      "               width: size_t, height: size_t, ambient: float }   (all mandatory)\n"
      "  vec3 = array of exactly three numbers\n",
      stdout);
  fflush(stdout);
}

}  // extern "C"
