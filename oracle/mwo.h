/* ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C, single-threaded CPU restatement of the reference's step+render hot path
 * (Farama-Foundation/Miniworld v2.1.0).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (miniworld_amd/,
 * csrc/) must never include, link or call anything in oracle/.
 *
 * PARITY STATUS
 *   dynamics  (mwo_dyn.c)    : pinned — checked against trajectories produced by the
 *                              reference's own miniworld.py/entity.py/math.py run under
 *                              GL stubs (tests/golden/ npz files, tools/gen_golden.py).
 *   pixels    (mwo_render.c) : PARITY UNPINNED — the reference's pixels come from a
 *                              third-party OpenGL driver (pyglet>=1.5.27,<2.0 -> libGL,
 *                              libGLU) that is neither vendored in the reference nor
 *                              installable here, and its tests hold no golden images
 *                              (tests/test_miniworld.py only checks 0<mean<255).  This
 *                              file restates the OpenGL 2.1 fixed-function semantics of
 *                              the call sites cited per function, with every
 *                              implementation-defined choice written down.
 *                              External anchor: the reference's own screenshots, compared per surface
 *                              in tests/test_oracle_vs_reference_screenshots.py (oracle/README.md).
 */
#ifndef MWO_H
#define MWO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene description (what the reference hands to GL, as data) -------------- */

typedef struct {            /* one polygon of display list 1: a room polygon (miniworld.py:401-434,
                             * Room._render) or a quad of a static ImageFrame / TextFrame
                             * (entity.py:193-259, 303-383) in world coordinates              */
    float v[4][3];          /* glVertex3f   */
    float uv[4][2];         /* glTexCoord2f */
    float n[3];             /* glNormal3f   */
    int32_t nv;             /* 3 or 4 vertices; | MWO_POLY_ENTITY for a quad of a static entity */
    int32_t tex;            /* index into scene.tex, or -1 (untextured) */
    float rgb[3];           /* glColor3f: 1,1,1 for rooms and frame fronts, 0,0,0 for frame borders */
    float xf[4];            /* MWO_POLY_XF: glTranslatef(x, y, z), glRotatef(angle, 0, 1, 0) in front of the quad
                             * (entity.py:205-207, 318-320); v, n are then the object-space glVertex3f / glNormal3f */
} mwo_poly;

#define MWO_POLY_ENTITY 0x100   /* a quad of a static ImageFrame / TextFrame                           */
#define MWO_POLY_XF     0x200   /* drawn under its own model transform (xf)                             */
#define MWO_POLY_QUAD   0x400   /* issued inside glBegin(GL_QUADS) (walls, frames); otherwise GL_POLYGON */

/* FRAME: an ImageFrame / TextFrame; its quads are in the polygon list, it is an entity only for
 * collisions (radius 0) and for get_visible_ents */
enum { MWO_ENT_NONE = 0, MWO_ENT_BOX = 1, MWO_ENT_MESH = 2, MWO_ENT_FRAME = 3 };

typedef struct {            /* entity.py:409-432 (Box.render), :150-161 (MeshEnt.render) */
    int32_t kind;
    int32_t mesh;           /* index into scene.meshes (kind == MESH) */
    double pos[3];
    double dir;             /* radians */
    double size[3];         /* Box: sx, sy, sz */
    double color[3];        /* Box: color_vec (already clipped to [0,1]) */
    double scale;           /* MeshEnt.scale */
    int32_t is_static;      /* drawn inside display list 1 (miniworld.py:1058-1060) rather than in immediate mode */
    int32_t pad;
} mwo_ent;

typedef struct {            /* opengl.py:148-184 (Texture.load): RGB8, rows bottom-up   */
    int32_t w, h, nlevels, pad;
    const uint8_t *rgb;     /* level 0..nlevels-1 concatenated */
} mwo_tex;

typedef struct {            /* objmesh.py:139-170: per-face-vertex arrays               */
    int32_t ntris;
    int32_t tex;            /* -1: untextured */
    const float *pos;       /* [ntris][3][3] */
    const float *nrm;       /* [ntris][3][3] */
    const float *uv;        /* [ntris][3][2] */
    const float *rgb;       /* [ntris][3][3]  (material Kd per vertex) */
} mwo_mesh;

typedef struct {
    int32_t width, height;  /* 80 x 60 */
    int32_t nsamples;       /* 8 (miniworld.py:515) */
    int32_t pad;
    double agent_pos[3];
    double agent_dir;
    double cam_height, cam_fwd_disp, cam_pitch, cam_fov_y;   /* entity.py:460-467 */
    double sky[3], light_pos[3], light_color[3], light_ambient[3];
    int32_t n_polys, n_ents, n_tex, n_mesh;
    const mwo_poly *polys;
    const mwo_ent *ents;    /* in draw order: static first, then dynamic (miniworld.py:1058-1077) */
    const mwo_tex *tex;
    const mwo_mesh *meshes;
    /* render_top_view (miniworld.py:1088-1175): view = 1 selects the orthographic map view over the
     * world extents (+-1 m, aspect-fitted); render_agent draws Agent.render's marker (entity.py:518-539) */
    int32_t view, render_agent;
    double extent[4];       /* env.min_x, max_x, min_z, max_z (miniworld.py:588-591) */
    double agent_radius, agent_height;
} mwo_scene;

/* ---- math --------------------------------------------------------------------- */
void mwo_sincos(double x, double *s, double *c);

/* ---- textures ----------------------------------------------------------------- */
/* Number of bytes of the full RGB8 pyramid of a w x h texture; *nlevels receives the count. */
int64_t mwo_mip_bytes(int32_t w, int32_t h, int32_t *nlevels);
/* Builds the pyramid (level 0 = a copy of rgb) into out. */
void mwo_build_mips(const uint8_t *rgb, int32_t w, int32_t h, uint8_t *out);

/* ---- render (render_obs + FrameBuffer.resolve + get_depth_map) ---------------- */
/* rgb   : uint8 [H][W][3], row 0 = top           (opengl.py:339-398)
 * z16   : uint16[H][W]    resolved depth buffer  (opengl.py:361-372), may be NULL
 * depth : float [H][W]    metres                 (opengl.py:400-435), may be NULL
 * prim  : int32 [H][W][nsamples] winning draw index per sample (-1 = sky), may be NULL */
int mwo_render_obs(const mwo_scene *sc, uint8_t *rgb, uint16_t *z16, float *depth, int32_t *prim);

/* MiniWorldEnv.get_visible_ents (miniworld.py:1238-1333).  sc->ents in self.entities order;
 * vis: uint8[n_ents], 1 = the entity's 0.2 m proxy box passed at least one sample. */
int mwo_visible_ents(const mwo_scene *sc, uint8_t *vis);

/* ---- dynamics (MiniWorldEnv.step and friends) --------------------------------- */
enum { MWO_TASK_NONE = 0, MWO_TASK_GOTO = 1, MWO_TASK_PICKUP = 2, MWO_TASK_PUTNEXT = 3 };

typedef struct {
    /* agent */
    double pos[3];
    double dir;
    double radius;          /* 0.4 (entity.py:470) */
    double cam_height;      /* used by _get_carry_pos (miniworld.py:615) */
    int32_t carrying;       /* entity index or -1 */
    int32_t step_count;
    /* episode constants */
    int32_t max_episode_steps;
    int32_t task;           /* MWO_TASK_* */
    int32_t goal_ent;       /* GOTO: entity index of the box */
    int32_t num_objs;       /* PICKUP */
    int32_t num_picked_up;
    int32_t n_ents;
    double max_forward_step;/* params.get_max("forward_step") (miniworld.py:581) */
    int32_t goal_ent2;      /* PUTNEXT: the entity goal_ent has to be put next to (putnext.py:74-78) */
    int32_t pad;
} mwo_agent_state;

typedef struct {            /* physical part of an entity (miniworld.py:951-961) */
    double pos[3];
    double dir;
    double radius;
    double height;
    int32_t alive;          /* 0 after removal from self.entities (pickupobjects.py:87) */
    int32_t is_static;
} mwo_phys_ent;

/* One MiniWorldEnv.step(action) + the env subclass' reward/termination rule
 * (miniworld.py:670-730; hallway.py:67-74; pickupobjects.py:83-95).
 * fwd_step/fwd_drift/turn_step are the three per-step params (miniworld.py:678-680).
 * segs: [n_segs][2][2] = (x,z) of both endpoints (miniworld.py:324-325).
 * Rendering happens between the physics and the task rule in the reference (:717);
 * this function therefore returns, in ents_at_render, the entity table as it must be
 * rendered (before PickupObjects removes the carried entity).  */
int mwo_step(mwo_agent_state *ag, mwo_phys_ent *ents, mwo_phys_ent *ents_at_render,
             const double *segs, int32_t n_segs, int32_t action,
             double fwd_step, double fwd_drift, double turn_step,
             double *reward, int32_t *terminated, int32_t *truncated);

/* MiniWorldEnv.intersect (miniworld.py:937-963).  self_idx = -1 tests on behalf of the
 * agent, otherwise on behalf of ents[self_idx] (then the agent's own circle is tested
 * too, as the agent is a member of self.entities, :907).
 * Returns 0 = nothing, -1 = wall (the reference's True), 1+i = ents[i], 1+n_ents = agent. */
int mwo_intersect(const mwo_agent_state *ag, const mwo_phys_ent *ents, int32_t self_idx,
                  double px, double pz, double radius, const double *segs, int32_t n_segs);

/* bench.py cpu_baseline: timed loop of mwo_step + mwo_render_obs on one env; returns seconds */
double mwo_bench_loop(mwo_scene *sc, mwo_agent_state *ag, mwo_phys_ent *ents, const double *segs,
                      int32_t n_segs, int32_t n_actions, int32_t steps, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
