/*
 * ctr_oracle.c — CPU restatement (plain C) of cutrace's per-pixel ray-cast +
 * shading path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline — never
 * as the product path.  The product (cutrace_amd/) does not link, import or
 * call anything in oracle/.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root).  Operation ORDER follows the reference exactly; build
 * with `-O2 -ffp-contract=off` and without -ffast-math so every float op is
 * rounded once, as in the reference headers compiled for the host with the
 * same flags (oracle/_ref, see oracle/Makefile).  Pinning: this restatement is
 * checked bit-for-bit against oracle/_ref (the reference's own headers
 * compiled in place) by tests/test_oracle_golden.py and against the committed
 * golden buffers under tests/golden/ that were generated from oracle/_ref.
 *
 * min/max semantics: the reference calls unqualified min()/max(); in the
 * host-compiled reference these bind to std::min/std::max
 * (a<b-style selects, NOT fminf/fmaxf), and that is what is restated here.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cutrace_amd.h"

typedef ctr_vec3 vec;

/* ---- inc/vector.hpp ------------------------------------------------------ */
static inline vec v3(float x, float y, float z) { vec r = {x, y, z}; return r; }
/* vector.hpp:99-101 */
static inline vec vadd(vec a, vec b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
/* vector.hpp:108-110 */
static inline vec vsub(vec a, vec b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
/* vector.hpp:117-119 (v*f and f*v are the same function: {f*x, f*y, f*z}) */
static inline vec vscale(vec a, float f) { return v3(f * a.x, f * a.y, f * a.z); }
/* vector.hpp:135-137 */
static inline vec vmul(vec a, vec b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* vector.hpp:126-128 */
static inline float vdot(vec a, vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vector.hpp:65-71 */
static inline vec vcross(vec a, vec o) {
  return v3(a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x);
}
/* vector.hpp:85-92 */
static inline float vnorm(vec a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* vector.hpp:77-79 */
static inline vec vnormalized(vec a) { return vscale(a, 1.0f / vnorm(a)); }
/* vector.hpp:204-206: incoming - 2.0f * (normal.dot(incoming)) * normal
 * parses as incoming - ((2.0f * dot) * normal) */
static inline vec vreflect(vec incoming, vec normal) {
  return vsub(incoming, vscale(normal, 2.0f * vdot(normal, incoming)));
}
/* vector.hpp:218-224, columns c0,c1,c2 */
static inline float det3(vec c0, vec c1, vec c2) {
  float a = c0.x, b = c1.x, c = c2.x, d = c0.y, e = c1.y, f = c2.y, g = c0.z, h = c1.z, i = c2.z;
  return a * e * i + b * f * g + c * d * h - c * e * g - a * f * h - b * d * i;
}
/* std::min / std::max as the host-compiled reference binds them */
static inline float smin(float a, float b) { return (b < a) ? b : a; }
static inline float smax(float a, float b) { return (a < b) ? b : a; }
static inline float comp(vec v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; } /* vector.hpp:37-39 */

typedef struct { vec start, dir; } ray; /* gpu_types.hpp ray */

typedef struct {
  const ctr_scene_desc *s;
  uint64_t casts;       /* ray_cast invocations */
  uint64_t alg_bytes;   /* SURVEY §8(d) algorithmic bytes of those casts */
  int trace;            /* debugging aid: print every cast (orc_trace_pixel) */
  int64_t last_tri;     /* index (within the mesh) of the triangle that won the last mesh_intersect */
  float *uv;            /* optional output: texture coordinates of the primary hit, 2 floats per pixel */
  int ignore_transparent_primary; /* the kernel.hpp:52 cast is made with ignore_transparent = true (orc_render_ex) */
} octx;

/* ---- inc/default_schema.hpp primitives ------------------------------------ */

/* triangle::intersect, default_schema.hpp:57-78 (uv_for :37-46 feeds only the
 * texture coordinates, which the solid material ignores :326-340; omitted). */
static int tri_intersect(vec p1, vec p2, vec p3, const ray *r, float min_t, vec *hit, float *dist, vec *normal) {
  vec a = vsub(p2, p1), b = vsub(p2, p3), c = r->dir, d = vsub(p2, r->start);
  float alpha = det3(a, b, c);            /* B  = {a,b,c} */
  float beta = det3(d, b, c) / alpha;     /* A1 = {d,b,c} */
  float gamma = det3(a, d, c) / alpha;    /* A2 = {a,d,c} */
  float t0 = det3(a, b, d) / alpha;       /* A  = {a,b,d} */
  if (beta >= 0 && gamma >= 0 && beta + gamma <= 1 && isfinite(t0) && min_t <= t0) {
    *dist = t0;
    *hit = vadd(r->start, vscale(r->dir, *dist));
    *normal = vscale(vnormalized(vcross(vsub(p2, p3), vsub(p1, p3))), -1.0f);
    return 1;
  }
  return 0;
}

/* mesh::bound_intersects, default_schema.hpp:99-114 */
static int bound_intersects(vec bmin, vec bmax, const ray *r) {
  float tmin = 0.0, tmax = INFINITY;
  vec r_inv = v3(1.0f / r->dir.x, 1.0f / r->dir.y, 1.0f / r->dir.z);
  for (int d = 0; d < 3; ++d) {
    float t1 = (comp(bmin, d) - comp(r->start, d)) * comp(r_inv, d);
    float t2 = (comp(bmax, d) - comp(r->start, d)) * comp(r_inv, d);
    tmin = smin(smax(t1, tmin), smax(t2, tmin));
    tmax = smax(smin(t1, tmax), smin(t2, tmax));
  }
  return tmin <= tmax;
}

/* mesh::intersect, default_schema.hpp:125-144 */
static int mesh_intersect(octx *cx, const ctr_object *o, const ray *r, float min_t, vec *hit, float *dist, vec *normal) {
  if (!bound_intersects(o->v0, o->v1, r)) return 0;
  cx->alg_bytes += 48ull * o->tri_count;
  vec h = {0, 0, 0}, n = {0, 0, 0};
  float t;
  *dist = INFINITY;
  const ctr_triangle *tris = cx->s->triangles + o->tri_begin;
  for (uint64_t k = 0; k < o->tri_count; k++) {
    if (tri_intersect(tris[k].p1, tris[k].p2, tris[k].p3, r, min_t, &h, &t, &n) && t < *dist) {
      *dist = t;
      *hit = h;
      *normal = n;
      cx->last_tri = (int64_t)k;
    }
  }
  return *dist != INFINITY;
}

/* plane::intersect, default_schema.hpp:189-201 (uv_for :169-178 omitted, unused) */
static int plane_intersect(const ctr_object *o, const ray *r, float min_t, vec *hit, float *dist, vec *n) {
  float t0 = vdot(vsub(o->v0, r->start), o->v1) / vdot(r->dir, o->v1);
  if (isfinite(t0) && min_t <= t0) {
    *dist = t0;
    *hit = vadd(r->start, vscale(r->dir, t0));
    *n = o->v1;
    return 1;
  }
  return 0;
}

/* sphere::intersect, default_schema.hpp:226-251 (atan2/asin uv :247-249 omitted, unused) */
static int sphere_intersect(const ctr_object *o, const ray *r, float min_t, vec *hit, float *dist, vec *normal) {
  vec d = vnormalized(r->dir), c = o->v0, e = r->start;
  float R = o->f0;
  float dec = -vdot(d, vsub(e, c));
  float sub = dec * dec - vdot(d, d) * (vdot(vsub(e, c), vsub(e, c)) - R * R);
  float t0 = (dec - sqrtf(sub)) / vdot(d, d), t1 = (dec + sqrtf(sub)) / vdot(d, d);
  int t0v = isfinite(t0) && min_t <= t0, t1v = isfinite(t1) && min_t <= t1;
  int condition = (t0v ? 2 : 0) + (t1v ? 1 : 0);
  switch (condition) {
    case 0: return 0;
    case 1: *dist = t1; break;
    case 2: *dist = t0; break;
    case 3: *dist = smin(t0, t1); break;
    default: break;
  }
  *hit = vadd(r->start, vscale(vnormalized(r->dir), *dist));
  *normal = vnormalized(vsub(*hit, c));
  return 1;
}

/* get_intersect visitor dispatch, gpu_types.hpp:75-81 / gpu_variant.hpp:216-240 */
static int obj_intersect(octx *cx, const ctr_object *o, const ray *r, float min_t, vec *hit, float *dist, vec *normal) {
  switch (o->type) {
    case CTR_OBJ_TRIANGLE: return tri_intersect(o->v0, o->v1, o->v2, r, min_t, hit, dist, normal);
    case CTR_OBJ_MESH: return mesh_intersect(cx, o, r, min_t, hit, dist, normal);
    case CTR_OBJ_PLANE: return plane_intersect(o, r, min_t, hit, dist, normal);
    default: return sphere_intersect(o, r, min_t, hit, dist, normal);
  }
}

/* ---- inc/ray_cast.hpp:29-55 ------------------------------------------------ */
static int ray_cast(octx *cx, const ray *finder, float min_dist, float *distance, uint64_t *hit_id, vec *hit_point, vec *normal, int ignore_transparent) {
  const ctr_scene_desc *s = cx->s;
  cx->casts++;
  cx->alg_bytes += 56ull * s->n_objects;
  *distance = INFINITY;
  vec hit = {0, 0, 0}, nrm = {0, 0, 0};
  float dist;
  int was_hit = 0;
  for (uint64_t i = 0; i < s->n_objects; i++) {
    /* ray_cast.hpp:39-40 (false at every call site of the reference; orc_render_ex can make the kernel.hpp:52 cast with
     * true): material::is_transparent, default_schema.hpp:334 — `transparency >= 1e-6`, a double comparison */
    if (ignore_transparent && (double)s->materials[s->objects[i].mat_idx].transparency >= 1e-6) continue;
    if (obj_intersect(cx, &s->objects[i], finder, min_dist, &hit, &dist, &nrm)) {
      if (dist > min_dist && dist < *distance) {
        *distance = dist;
        *hit_id = i;
        *hit_point = hit;
        *normal = nrm;
        was_hit = 1;
      }
    }
  }
  if (cx->trace) {
    printf("cast o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) min=%.9g -> hit=%d obj=%lld tri=%lld t=%.9g\n", finder->start.x,
           finder->start.y, finder->start.z, finder->dir.x, finder->dir.y, finder->dir.z, min_dist, was_hit,
           was_hit ? (long long)*hit_id : -1LL,
           (was_hit && s->objects[*hit_id].type == CTR_OBJ_MESH) ? (long long)cx->last_tri : -1LL, *distance);
  }
  return was_hit;
}

/* ---- inc/shading.hpp:22-45 -------------------------------------------------- */
static float shadow_intensity(octx *cx, const ray *shadow_ray, float max_dist) {
  const ctr_scene_desc *s = cx->s;
  float intensity = 0.0f;
  float last_hit = 0.0f;
  ray check = {shadow_ray->start, shadow_ray->dir};
  float dist;
  uint64_t h = 0;
  vec hit = {0, 0, 0}, normal = {0, 0, 0};
  /* shading.hpp:32: `last_hit + 1e-3` is a DOUBLE add narrowed to float at the call */
  while (ray_cast(cx, &check, (float)((double)last_hit + 1e-3), &dist, &h, &hit, &normal, 0) && dist < max_dist) {
    float trans = s->materials[s->objects[h].mat_idx].transparency; /* get_bounce_params, default_schema.hpp:337-340 */
    intensity += (1.0f - trans);
    if (intensity >= 1.0f) return 1.0f;
    last_hit = dist;
  }
  return intensity;
}

/* ---- inc/shading.hpp:64-99 -------------------------------------------------- */
static vec phong(octx *cx, const ray *incoming, const vec *hit, uint64_t hit_id, const vec *normal, float ambient) {
  const ctr_scene_desc *s = cx->s;
  const ctr_material *m = &s->materials[s->objects[hit_id].mat_idx];
  /* get_phong_params, default_schema.hpp:326-332 */
  vec diffuse = m->color, specular = vscale(m->color, m->specular);
  float phong_exp = m->phong_exp;
  vec final = vscale(diffuse, ambient);
  vec direction = {0, 0, 0};
  float distance = INFINITY;
  for (uint64_t li = 0; li < s->n_lights; li++) {
    const ctr_light *l = &s->lights[li];
    if (l->type == CTR_LIGHT_SUN) { /* sun::direction_to, default_schema.hpp:280-283 */
      direction = vscale(l->v, -1.0f);
      distance = INFINITY;
    } else { /* point_light::direction_to, default_schema.hpp:305-308 */
      direction = vnormalized(vsub(l->v, *hit));
      distance = vnorm(vsub(l->v, *hit));
    }
    ray shadow = {*hit, vnormalized(direction)};
    float light_dist = distance * vnorm(direction);
    vec color = l->color;
    vec nn = vnormalized(*normal), nd = vnormalized(direction);
    float shadow_fac = shadow_intensity(cx, &shadow, light_dist);
    if (shadow_fac < 1.0f) {
      float fd = smax(0.0f, vdot(nn, nd));
      vec ld = vmul(diffuse, color);
      vec h = vnormalized(vadd(vscale(vnormalized(incoming->dir), -1.0f), nd));
      float fs = powf(smax(0.0f, vdot(nn, h)), phong_exp);
      vec ls = vmul(specular, color);
      /* final += (1 - shadow_fac) * (fd * ld + fs * ls) */
      vec term = vscale(vadd(vscale(ld, fd), vscale(ls, fs)), 1 - shadow_fac);
      final = vadd(final, term);
    }
  }
  return final;
}

/* ---- inc/shading.hpp:116-154 (template recursion → runtime recursion) ------- */
static vec ray_color(octx *cx, const ray *incoming, float min_t, float ambient, int bounces) {
  const ctr_scene_desc *s = cx->s;
  uint64_t id = 0;
  vec normal = {0, 0, 0}, rgb = {0.0f, 0.0f, 0.0f}, hit = {0, 0, 0};
  float distance;
  if (ray_cast(cx, incoming, min_t, &distance, &id, &hit, &normal, 0)) {
    rgb = phong(cx, incoming, &hit, id, &normal, ambient);
    if (bounces != 0) {
      const ctr_material *m = &s->materials[s->objects[id].mat_idx];
      float reflective = m->reflexivity, translucent = m->transparency;
      if (reflective >= 1e-6) { /* float vs double literal: compared in double */
        vec nd = vnormalized(incoming->dir), nn = vnormalized(normal);
        ray reflection = {vadd(incoming->start, vscale(incoming->dir, distance)), vreflect(nd, nn)};
        vec r_rgb = ray_color(cx, &reflection, min_t, ambient, bounces - 1);
        rgb = vadd(rgb, vscale(r_rgb, reflective));
      }
      if (translucent >= 1e-6) {
        ray pass = {vadd(incoming->start, vscale(incoming->dir, distance)), incoming->dir};
        vec t_rgb = ray_color(cx, &pass, min_t, ambient, bounces - 1);
        rgb = vadd(vscale(rgb, 1.0f - translucent), vscale(t_rgb, translucent));
      }
    }
  }
  return rgb;
}

/* cam::get_ray, default_schema.hpp:376-386 */
static ray cam_get_ray(const ctr_camera *cam, uint64_t x, uint64_t y) {
  float aspect = (float)cam->w / (float)cam->h;
  vec x_v = vscale(cam->right, (((float)x / (float)cam->w) - 0.5f) * aspect);
  vec y_v = vscale(cam->up, 0.5f - ((float)y / (float)cam->h));
  vec z_v = cam->forward;
  ray r = {cam->pos, vnormalized(vadd(vadd(x_v, y_v), z_v))};
  return r;
}

/* Texture coordinates of a hit (ray_cast's tex_coords, ray_cast.hpp:47).  Every primitive computes them from the hit
 * point alone, so they are restated as a function of the winning object and its hit point:
 *   triangle::uv_for  default_schema.hpp:37-46     plane::uv_for  :169-178
 *   sphere            :246-249 (delta = (hit - center).normalized(), atan2 / asin)
 *   mesh              :138-139 (u = hit.x, v = hit.y)
 * A miss leaves the value-initialised uv{} of kernel.hpp:51: (0, 0). */
static void uv_of_hit(const ctr_object *o, vec hit, float *u, float *v) {
  switch (o->type) {
    case CTR_OBJ_TRIANGLE: {
      vec p2p1 = vsub(o->v1, o->v0), p3p1 = vsub(o->v2, o->v0), xp1 = vsub(hit, o->v0);
      vec proj_u = vscale(p2p1, vdot(xp1, p2p1) / vdot(p2p1, p2p1));
      vec proj_v = vscale(p3p1, vdot(xp1, p3p1) / vdot(p3p1, p3p1));
      *u = vnorm(proj_u) / vnorm(p2p1);
      *v = vnorm(proj_v) / vnorm(p3p1);
      break;
    }
    case CTR_OBJ_MESH: *u = hit.x; *v = hit.y; break;
    case CTR_OBJ_PLANE: {
      vec normal = o->v1;
      vec ax1 = vnormalized(v3(normal.y, -normal.x, 0.0f));
      vec ax2 = vcross(normal, ax1);
      vec mod_pt = vsub(o->v0, hit);
      *u = vdot(ax1, mod_pt);
      *v = vdot(ax2, mod_pt);
      break;
    }
    default: {
      vec delta = vnormalized(vsub(hit, o->v0));
      *u = 0.5f + (atan2f(delta.z, delta.x) / (2.0f * (float)M_PI));
      *v = 0.5f + (asinf(delta.y) / (float)M_PI);
      break;
    }
  }
}

/* render_kernel body for one pixel, kernel.hpp:44-59 */
static void render_pixel(octx *cx, uint64_t x_id, uint64_t y_id, float fudge, int bounces, float *depth, vec *color, vec *normals, int64_t *hit_ids, uint64_t out_idx) {
  const ctr_scene_desc *s = cx->s;
  float dist = INFINITY;
  ray r = cam_get_ray(&s->cam, x_id, y_id);
  uint64_t hit_id = s->n_objects;
  vec hit_point = {0, 0, 0}, normal = {0, 0, 0};
  int did_hit = ray_cast(cx, &r, fudge, &dist, &hit_id, &hit_point, &normal, cx->ignore_transparent_primary);
  depth[out_idx] = dist;
  normals[out_idx] = normal;
  if (hit_ids) hit_ids[out_idx] = did_hit ? (int64_t)hit_id : -1;
  if (cx->uv) {
    float u = 0.0f, v = 0.0f;
    if (did_hit) uv_of_hit(&s->objects[hit_id], hit_point, &u, &v);
    cx->uv[2 * out_idx + 0] = u;
    cx->uv[2 * out_idx + 1] = v;
  }
  color[out_idx] = ray_color(cx, &r, fudge, s->cam.ambient, bounces);
}

/* ---- driver: rows handed to pthread workers ------------------------------------ */
typedef struct {
  const ctr_scene_desc *s;
  float fudge;
  int bounces;
  const uint64_t *rows;  /* selected global rows */
  uint64_t n_rows;
  float *depth;
  vec *color, *normals;
  int64_t *hit_ids;
  uint64_t *next;        /* shared atomic row counter */
  uint64_t casts, alg_bytes;
  float *uv;
  int ign;
} job;

static void *worker(void *arg) {
  job *j = (job *)arg;
  octx cx = {j->s, 0, 0, 0, -1, j->uv, j->ign};
  uint64_t w = j->s->cam.w;
  for (;;) {
    uint64_t k = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
    if (k >= j->n_rows) break;
    uint64_t y = j->rows[k];
    for (uint64_t x = 0; x < w; x++) render_pixel(&cx, x, y, j->fudge, j->bounces, j->depth, j->color, j->normals, j->hit_ids, k * w + x);
  }
  j->casts = cx.casts;
  j->alg_bytes = cx.alg_bytes;
  return 0;
}

static int row_selected(const ctr_rows *r, uint64_t y) {
  return y >= r->row_begin && y < r->row_end && ((y / r->block_rows) % r->n_parts) == r->part;
}

/*
 * Oracle entry point.  Same buffers/semantics as ctr_render (host-buffer form).
 * hit_ids (optional): object index of the primary hit, -1 on miss.
 * counters (optional): [0]=ray_cast invocations, [1]=algorithmic bytes
 * (56·n_obj per cast + 48·n_tri per AABB-hit mesh + 28 per pixel).
 */
int orc_render_uv(const ctr_scene_desc *s, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2);
int orc_render(const ctr_scene_desc *s, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
               float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters) {
  return orc_render_uv(s, fudge, bounces, rows_in, n_threads, depth, color3, normal3, hit_ids, counters, 0);
}

/* ... plus uv2 (optional): texture coordinates of the primary hit, 2 floats per pixel (0, 0 on a miss) */
int orc_render_ex(const ctr_scene_desc *s, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2, int ignore_transparent_primary);
int orc_render_uv(const ctr_scene_desc *s, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2) {
  return orc_render_ex(s, fudge, bounces, rows_in, n_threads, depth, color3, normal3, hit_ids, counters, uv2, 0);
}

/* ... plus ignore_transparent_primary: the cast of kernel.hpp:52 (depth, normal, hit id, uv) is made with ray_cast's
 * ignore_transparent = true (ray_cast.hpp:30,39-40) — objects whose material is transparent do not exist for it; ray_color's
 * own casts stay as the reference's shading code makes them (shading.hpp:32,123 pass false) */
int orc_render_ex(const ctr_scene_desc *s, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2, int ignore_transparent_primary) {
  ctr_rows rr = {0, s->cam.h, s->cam.h ? s->cam.h : 1, 0, 1};
  if (rows_in && rows_in->row_end > rows_in->row_begin) {
    rr = *rows_in;
    if (rr.block_rows == 0) rr.block_rows = s->cam.h ? s->cam.h : 1;
    if (rr.n_parts == 0) { rr.n_parts = 1; rr.part = 0; }
    if (rr.row_end > s->cam.h) rr.row_end = s->cam.h;
  }
  uint64_t *sel = (uint64_t *)malloc(sizeof(uint64_t) * (s->cam.h + 1));
  uint64_t n = 0;
  for (uint64_t y = 0; y < s->cam.h; y++) if (row_selected(&rr, y)) sel[n++] = y;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  uint64_t next = 0;
  job *jobs = (job *)calloc((size_t)n_threads, sizeof(job));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int t = 0; t < n_threads; t++) {
    job jj = {s, fudge, bounces, sel, n, depth, (vec *)color3, (vec *)normal3, hit_ids, &next, 0, 0, uv2, ignore_transparent_primary};
    jobs[t] = jj;
  }
  if (n_threads == 1) worker(&jobs[0]);
  else {
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], 0, worker, &jobs[t]);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], 0);
  }
  uint64_t casts = 0, bytes = 0;
  for (int t = 0; t < n_threads; t++) { casts += jobs[t].casts; bytes += jobs[t].alg_bytes; }
  if (counters) { counters[0] = casts; counters[1] = bytes + 28ull * n * s->cam.w; }
  free(jobs); free(th); free(sel);
  return 0;
}

/* debugging aid: print every ray_cast of one pixel (origin, direction, winner) */
void orc_trace_pixel(const ctr_scene_desc *s, uint64_t x, uint64_t y, float fudge, int bounces) {
  octx cx = {s, 0, 0, 1, -1, 0};
  float depth;
  vec color, normal;
  render_pixel(&cx, x, y, fudge, bounces, &depth, &color, &normal, 0, 0);
  printf("pixel (%llu,%llu): depth=%.9g color=(%.9g %.9g %.9g) casts=%llu\n", (unsigned long long)x,
         (unsigned long long)y, depth, color.x, color.y, color.z, (unsigned long long)cx.casts);
}

/* cam::look_at restated for the oracle's own use (default_schema.hpp:370-374) */
void orc_look_at(ctr_camera *cam, ctr_vec3 eye, ctr_vec3 up_hint, ctr_vec3 look) {
  cam->pos = eye;
  cam->up = up_hint;
  cam->forward = vnormalized(vsub(look, cam->pos));
  cam->right = vnormalized(vcross(cam->forward, cam->up));
  cam->up = vnormalized(vcross(cam->right, cam->forward));
}

/* images.hpp:27-29, 48-54, 73-76 — float → u8 quantisation of the three maps
 * (checker for the image-writer row). out: 3 bytes / pixel. */
void orc_quantise_depth(const float *depth, uint64_t n, float max_d, unsigned char *out) {
  for (uint64_t i = 0; i < n; i++) {
    float v = depth[i];
    unsigned char b = isfinite(v) ? (unsigned char)(255 * (max_d - v) / max_d) : 0;
    out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = b;
  }
}
void orc_quantise_normal(const float *n3, uint64_t n, unsigned char *out) {
  for (uint64_t i = 0; i < n; i++) {
    vec v = v3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
    if (vnorm(v) <= 1e-6) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0; continue; }
    vec q = vadd(v3(0.5f, 0.5f, 0.5f), vscale(vnormalized(v), 0.5f));
    out[3 * i] = (unsigned char)(255 * q.x);
    out[3 * i + 1] = (unsigned char)(255 * q.y);
    out[3 * i + 2] = (unsigned char)(255 * q.z);
  }
}
void orc_quantise_color(const float *c3, uint64_t n, unsigned char *out) {
  for (uint64_t i = 0; i < 3 * n; i++) {
    float c = smin(1.0f, smax(0.0f, c3[i]));
    out[i] = (unsigned char)(255 * c);
  }
}
