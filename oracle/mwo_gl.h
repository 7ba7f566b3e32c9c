/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Internal interface between the two halves of the pixel oracle:
 *   mwo_geom.c    what the reference's GL calls do to a vertex before rasterisation (Mesa's matrix stack, the
 *                 fixed-function vertex program, gallium's draw module: decomposition, clipping, viewport)
 *   mwo_render.c  llvmpipe's triangle setup, rasteriser, fragment pipeline, texture sampler, resolve
 * Both restate Mesa 23.2.1 / llvmpipe (LLVM 15, x86-64 with FMA) as observed on the reference's own frames:
 * tools/refshim_gl.py runs /root/reference/miniworld unmodified on that driver, tests/golden/gl_*.npz hold its
 * frames, tests/test_oracle_vs_reference_gl.py compares.
 */
#ifndef MWO_GL_H
#define MWO_GL_H
#include "mwo.h"

typedef struct { float m[16]; } mwo_mat4;          /* column-major like GL: m[col * 4 + row] */

typedef struct {
    float clip[4];          /* clip coordinates                                                      */
    float win[4];           /* window x, y (GL frame-buffer space, y up), z in [0,1], 1 / w_clip      */
    float col[4];           /* lit vertex colour, clamped to [0,1]                                   */
    float st[2];            /* texture coordinates                                                   */
    unsigned clipmask;
} mwo_vert;

typedef struct {
    mwo_vert v[3];          /* in the order the rasteriser's setup receives them                     */
    int32_t tex;            /* texture index or -1                                                   */
    int32_t draw;           /* index of the GL primitive (polygon / quad / triangle) it came from    */
} mwo_tri;

typedef struct {
    mwo_tri *tris;
    int n, cap;
} mwo_trilist;

/* Runs the vertex half of render_obs (view 0), render_top_view (view 1) or get_visible_ents' proxy pass
 * (proxies = 1: rooms untextured, then one list of proxy-box triangles per entity, whose first-triangle
 * index is stored in ent_first[e], ent_first[n_ents] = total).  Returns 0 or a negative error. */
int mwo_geometry(const mwo_scene *sc, int proxies, mwo_trilist *out, int *ent_first);
void mwo_trilist_free(mwo_trilist *l);

/* glibc 2.35 sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, the FMA multiarch variant
 * that x86-64 machines with FMA select): Mesa's _math_matrix_rotate calls them. */
float mwo_sinf(float x);
float mwo_cosf(float x);

#endif
