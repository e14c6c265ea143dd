/*
 * cutrace_host.h — C-ABI of the host-side pieces either side of the hot path:
 * the scene-JSON loader (reference inc/loader.hpp:679-780 +
 * inc/default_schema.hpp:463-898), the binary-STL mesh reader (replaces the
 * Assimp call at inc/default_schema.hpp:516-545), and the image writers
 * (inc/images.hpp:26-88).  Pure CPU; no HIP dependency.
 */
#ifndef CUTRACE_HOST_H
#define CUTRACE_HOST_H

#include <stdint.h>
#include "cutrace_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host arithmetic feeding the hot path (same operations as the reference's host code) -- */
/* cam::look_at (default_schema.hpp:370-374): forward, right, up from eye/up/look */
void ctr_camera_look_at(ctr_camera *cam, ctr_vec3 eye, ctr_vec3 up_hint, ctr_vec3 look);
/* mesh::bounding_box (default_schema.hpp:573-586) */
void ctr_mesh_bounds(const ctr_triangle *tris, uint64_t n, ctr_vec3 *bb_min, ctr_vec3 *bb_max);
/* number of rows selected by a ctr_rows for an image of height h */
uint64_t ctr_rows_count(const ctr_rows *rows, uint64_t h);

typedef struct ctr_host_scene ctr_host_scene; /* owns the flat arrays behind a ctr_scene_desc */

/* default_schema::load_file (loader.hpp:763-780).  Diagnostics go to stderr with
 * the reference's wording.  Returns CTR_OK and a scene when every element
 * loaded (last_was_success == true), CTR_E_PARSE otherwise (*out is still set
 * to the partially loaded scene, as the reference keeps going), CTR_E_IO when
 * the file cannot be read.  mesh "file" paths are relative to the CWD
 * (schema.md:73-74). */
int ctr_host_scene_load(const char *json_path, ctr_host_scene **out);
/* same, from an in-memory JSON text */
int ctr_host_scene_parse(const char *json_text, ctr_host_scene **out);
void ctr_host_scene_free(ctr_host_scene *hs);
/* the flat description (valid until the host scene is freed or edited) */
const ctr_scene_desc *ctr_host_scene_desc(ctr_host_scene *hs);
/* bench / test overrides */
void ctr_host_scene_set_size(ctr_host_scene *hs, uint64_t w, uint64_t h);
int ctr_host_scene_set_material(ctr_host_scene *hs, uint64_t idx, const ctr_material *m);

/* binary STL → triangles in facet order, v1,v2,v3 → p1,p2,p3.
 * Returns the number of triangles written (≤ cap), or the total count when
 * tris == NULL; negative status on error. */
int64_t ctr_stl_read(const char *path, ctr_triangle *tris, uint64_t cap);
int ctr_stl_write(const char *path, const ctr_triangle *tris, uint64_t n);

/* dump_scene text (kernel.hpp:150-166) on stdout, from the flat scene */
void ctr_dump_scene(const ctr_scene_desc *desc);
/* the static schema description printed when a load fails (main.cu:16-19) */
void ctr_dump_schema(void);

/* images.hpp:26-88: float → u8 quantisation (3 bytes per pixel out) */
void ctr_quantise_depth(const float *depth, uint64_t n, float max_d, unsigned char *rgb_out);
void ctr_quantise_normal(const float *normal3, uint64_t n, unsigned char *rgb_out);
void ctr_quantise_color(const float *color3, uint64_t n, unsigned char *rgb_out);
/* baseline JPEG, 3 components, quality as stbi_write_jpg's (images.hpp:39,64,86 use 90) */
int ctr_write_jpg(const char *path, int w, int h, const unsigned char *rgb, int quality);
/* write_depth_map / write_normal_map / write_colorized (images.hpp:26,47,72) */
int ctr_write_depth_map(const char *path, const float *depth, uint64_t w, uint64_t h, float max_d);
int ctr_write_normal_map(const char *path, const float *normal3, uint64_t w, uint64_t h);
int ctr_write_colorized(const char *path, const float *color3, uint64_t w, uint64_t h);

#ifdef __cplusplus
}
#endif
#endif
