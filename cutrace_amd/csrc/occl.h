// occl.h — per-light occluder-distance maps (host builder; render_kernel.hip "occluder map").
//
// shadow_intensity (inc/shading.hpp:22-45) casts a ray from every shaded hit to every light.  Most of those rays meet no
// mesh triangle at all — on the 16-mesh frame 54 % of the shadow casts of a wave start on a wall and reach their light
// freely — and yet each walks the top-level tree, passes a mesh's AABB test and visits half a dozen BVH nodes to learn
// that.  For a POINT light all those rays converge in one point, so "can anything be in the way?" has a cheap conservative
// answer that does not depend on the receiver: a cube map around the light whose cell holds a LOWER BOUND of the distance
// from the light to any mesh triangle seen in that cell's directions.  A shadow ray whose receiver is nearer to the light than
// that bound meets no mesh triangle before the light: the cast leaves the meshes out (planes, spheres and stand-alone
// triangles are tested as before).  The reference's result is unchanged: a mesh contributes to a shadow cast only through a
// triangle hit at a distance in (min_t, light distance) (shading.hpp:32: `dist < max_dist`), and there is none.
//
// Conservative by construction: a triangle is entered with its smallest distance to the light (closest point, in double,
// lowered by 2^-10) into every cell its solid angle touches — clipped against each cube face's frustum widened by two cells,
// its bounding box of cells grown by one more — so neither the cell a direction falls into nor the rounding of the kernel's
// lookup can miss it.  The one regime in which the reference's float test reports a hit for a ray that passes far from a
// triangle — the ray lies in the triangle's plane, i.e. for a shadow ray the LIGHT does (ctr_api.cpp refresh_linear_meshes) —
// switches the light's map off (all zeros: nothing is ever nearer than 0).  Sun lights get a zero map too (their rays do not
// converge; the reference's distance to a sun is infinite, which no bound exceeds).
#ifndef CUTRACE_AMD_OCCL_H
#define CUTRACE_AMD_OCCL_H

#include <stdint.h>
#include <vector>

#define CTR_OCCL_RES 128u                                   /* cells per cube-face edge */
#define CTR_OCCL_CELLS (6u * CTR_OCCL_RES * CTR_OCCL_RES)   /* floats per light */

struct OcclTri { float p[3][3]; };  // a mesh triangle's corners

// map: CTR_OCCL_CELLS floats, cell (face, row, col) at (face * RES + row) * RES + col;
// face 0/1 = +x/-x (col <- y, row <- z), 2/3 = +y/-y (col <- x, row <- z), 4/5 = +z/-z (col <- x, row <- y), the
// minor components divided by |major| and mapped from [-1, 1] to [0, RES).  Returns false (and a zero map) when the light lies
// ON a triangle.  Thread-safe per call.
bool occl_build_point_light(const float light[3], const OcclTri *tris, uint64_t n_tris, float *map);

// the cell the kernel looks up for a direction v (light -> receiver): the same selection rule, for tests
uint32_t occl_cell_of(const float v[3]);

#endif
