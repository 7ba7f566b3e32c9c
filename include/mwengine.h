/* mwengine — C ABI of the MI355X-native batched Miniworld step+render engine.
 *
 * The reference (Farama-Foundation/Miniworld v2.1.0) has no FFI: its "backend" is the
 * Python module miniworld/opengl.py (Texture, FrameBuffer, drawBox) plus raw GL calls in
 * Entity.render(), driven once per MiniWorldEnv.step().  This header is the seam a
 * maintainer binds instead of that module (ctypes stub: INTEGRATION.md).  Every entry
 * point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no exceptions: every call returns 0 on success or a negative MW_E_* code;
 *     mw_last_error() gives the message (reference: Python assert / exception).
 *   - the CALLER owns all output buffers (device pointers, e.g. torch tensors); the engine
 *     owns the Structure-of-Arrays world state of its N environments.
 *   - all device work is enqueued on the hipStream_t passed as `stream` (void* so that this
 *     header needs no HIP include); nothing synchronises unless documented.
 *   - one engine per device; an engine is not re-entrant (the reference is single-threaded:
 *     one GL context per env, miniworld.py:1187), distinct engines are independent.
 */
#ifndef MWENGINE_H
#define MWENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MW_ABI_VERSION 4

enum {
    MW_OK = 0,
    MW_E_INVALID = -1,      /* bad argument                                 */
    MW_E_HIP = -2,          /* HIP runtime error (message has the hipError) */
    MW_E_NOMEM = -3,
    MW_E_CAPACITY = -4,     /* more polys / entities / textures than configured */
    MW_E_DEVICE = -5,       /* no usable gfx950 device                      */
    MW_E_OVERFLOW = -6      /* a kernel reported a per-env capacity overflow */
};

/* entity kinds (entity.py: Box :386, MeshEnt :124 / Ball :445 / Key :435, ImageFrame :168 / TextFrame :262).
 * MW_ENT_FRAME: the entity's quads are part of the static polygon list (mw_set_geometry) — it is an entity
 * only for collisions (radius 0, miniworld.py:951-961) and for mw_visible_ents. */
enum { MW_ENT_NONE = 0, MW_ENT_BOX = 1, MW_ENT_MESH = 2, MW_ENT_FRAME = 3 };

/* env reward / termination rule applied after MiniWorldEnv.step (miniworld.py:670-730) */
enum {
    MW_TASK_NONE = 0,
    MW_TASK_GOTO = 1,       /* hallway.py:67-74, oneroom.py:64-71, maze.py:155-162 */
    MW_TASK_PICKUP = 2,     /* pickupobjects.py:83-95                            */
    MW_TASK_PUTNEXT = 3,    /* putnext.py:71-80: goal_ent next to goal_ent2, not carrying */
    MW_TASK_SIDEWALK = 4,   /* sidewalk.py:93-104: the street ends the episode with reward 0, the box like GOTO
                             * (mw_gen_program.street, goal_ent) */
    MW_TASK_SIGN = 5,       /* sign.py:152-170: action move_forward + 1 ends the episode; touching an object of the
                             * table ends it with +-1, the last one touched wins (mw_gen_program.sign_*) */
    MW_TASK_COLLECT = 6     /* collecthealth.py:79-98: health -2 per step, +2 reward while alive, -100 and the end at 0;
                             * a kit picked up is consumed after the frame was drawn: health back to 100, the kit leaves
                             * the entity list and is re-placed at its END with the env's own stream (place_entity) */
};

/* device-side world generators for mw_reset / auto-reset (the env's _gen_world) */
enum {
    MW_GEN_NONE = 0,        /* worlds only come from mw_set_state             */
    MW_GEN_HALLWAY = 1,     /* hallway.py:55-65                               */
    MW_GEN_ONEROOM = 2,     /* oneroom.py:59-62                               */
    MW_GEN_PICKUP = 3,      /* pickupobjects.py:55-81                         */
    MW_GEN_MAZE = 4,        /* maze.py:73-153 (needs shared_geometry = 0)     */
    MW_GEN_PROGRAM = 5      /* a fixed floorplan whose _gen_world is a list of draws and placements: the placement
                             * program of mw_set_gen_program (FourRooms, TMaze*, YMaze*, WallGap, ThreeRooms, PutNext,
                             * RoomObjects, Sidewalk, Sign, ...) */
};

/* Random stream of device-side resets.  MW_RNG_PHILOX: Philox4x32-10 keyed by the env's seed (same
 * distributions as the reference, different numbers; every generator, with or without domain
 * randomisation).  MW_RNG_PCG64: numpy's Generator(PCG64(SeedSequence(seed))) itself, drawn in the
 * reference's call order (miniworld.py:551, 872-905; hallway.py:59-65, oneroom.py:61-62), so that env i
 * reset with seed s is the world of the reference's env.reset(seed=s), and later episodes continue that
 * stream like env.reset() does (with domain randomisation the per-step parameters come from it too,
 * miniworld.py:677-680).  Every device generator, with and without domain_rand (the Maze's room textures exist in one
 * variant each in the reference, so Room._gen_static_data's variant draws consume nothing there, opengl.py:134-138). */
enum { MW_RNG_PHILOX = 0, MW_RNG_PCG64 = 1 };
enum { MW_AUTORESET_OFF = 0, MW_AUTORESET_SAME_STEP = 1 };

typedef struct mw_engine mw_engine;

/* Scalar simulation parameter with its domain-randomisation range (params.py:7-130). */
typedef struct { double def, lo, hi; } mw_range;

typedef struct {
    int32_t abi_version;        /* MW_ABI_VERSION */
    int32_t device_id;
    int32_t num_envs;
    int32_t obs_width, obs_height;  /* MiniWorldEnv(obs_width=80, obs_height=60) miniworld.py:473-474 */
    int32_t msaa;               /* FrameBuffer(..., num_samples=8) miniworld.py:515.  8 = the hot path; 4 or 1 = what the
                                 * reference falls back to on a driver that clamps GL_MAX_SAMPLES (opengl.py:229-231):
                                 * same semantics through the generic-resolution kernels, plain HWC layout only       */
    int32_t max_ents;           /* entity slots per env, agent excluded            */
    int32_t max_polys;          /* room polygons per geometry set                  */
    int32_t max_segs;           /* collision segments per geometry set             */
    int32_t max_visible;        /* GL primitives (polygons, box faces) that can be in view per env: the triangle list holds 6 x this (two triangles per primitive, three pieces each after clipping — a heuristic: a triangle across the near plane and two side planes clips to four or five pieces, back-face culling halves the list; mw_get_list_lengths reports what a workload needs).  A longer list is an error of mw_check, never a write out of bounds */
    int32_t shared_geometry;    /* 1: one geometry set for all envs, 0: one per env */
    int32_t task;               /* MW_TASK_*                                       */
    int32_t goal_ent;           /* MW_TASK_GOTO: entity slot of the box            */
    int32_t goal_ent2;          /* MW_TASK_PUTNEXT: slot of the second entity      */
    int32_t num_objs;           /* MW_TASK_PICKUP                                  */
    int32_t max_episode_steps;  /* miniworld.py:472, per env class                 */
    int32_t domain_rand;        /* miniworld.py:478                                */
    int32_t generator;          /* MW_GEN_*                                        */
    int32_t autoreset;          /* MW_AUTORESET_*                                  */
    double agent_radius;        /* entity.py:470 (0.4)                             */
    double agent_height;        /* entity.py:471 (1.6): height of the top-view marker */
    double max_forward_step;    /* params.get_max("forward_step") miniworld.py:581 */
    mw_range forward_step, forward_drift, turn_step;   /* params.py:123-125, miniworld.py:678-680 */
    /* per-episode parameters sampled by reset (miniworld.py:576-585, entity.py:405-407, 505-515) */
    mw_range sky_color[3], light_pos[3], light_color[3], light_ambient[3];   /* params.py:116-121 */
    mw_range obj_color_bias[3];                                               /* params.py:122     */
    mw_range cam_height, cam_fwd_disp, cam_pitch, cam_fov_y;                  /* params.py:127-130 */
    /* generator parameters: [0..3] room min_x,max_x,min_z,max_z; [4] box min_x override;
     * [5] agent max_x override; [6] agent |dir| range; [7] box size */
    double gen_args[8];
    /* MW_GEN_PICKUP: per object kind (Ball, Box, Key — pickupobjects.py:65): radius, height, scale,
     * first mesh id (mesh ids of one kind are consecutive in sorted colour order); and the RGB of the
     * six colours in sorted name order (entity.py:30-43) */
    double gen_tab[12];
    double gen_colors[18];
    /* Texture domain randomisation for generated single-room worlds (Texture.get with an rng,
     * opengl.py:124-140; Room._gen_static_data :295-297): per slot (0 wall, 1 floor, 2 ceiling —
     * the reference's draw order of rng.integers) the number of variants, their texture ids and
     * TEX_DENSITY / size (u, v).  n = 0 disables (geometry stays the shared set). */
    int32_t tex_nvar[3];
    int32_t tex_var_id[3][9];
    double tex_var_scale[3][9][2];
    double room_wall_height;    /* Room.wall_height of the generated room (2.74) */
    int32_t room_no_ceiling;    /* Room(no_ceiling=True) */
    int32_t rng_mode;           /* MW_RNG_* stream of the device generators */
} mw_config;

#define MW_POLY_ENTITY 0x100     /* a quad of a static ImageFrame / TextFrame: not drawn by mw_visible_ents            */
#define MW_POLY_XF     0x200     /* drawn under its own model transform: glTranslatef(xf[0..2]), glRotatef(xf[3], 0, 1, 0) */
#define MW_POLY_QUAD   0x400     /* issued inside glBegin(GL_QUADS) (walls, frames); otherwise GL_POLYGON (floor, ceiling) */

/* One static polygon exactly as it is fed to GL inside display list 1: a room polygon of Room._render
 * (miniworld.py:401-434, colour 1,1,1; floor and ceiling are GL_POLYGONs, walls GL_QUADS — the driver splits the two
 * kinds into different triangle pairs) or a quad of a static ImageFrame / TextFrame (entity.py:193-259, 303-383:
 * textured front in 1,1,1, border in 0,0,0) in the frame's OBJECT space with the arguments of the glTranslatef /
 * glRotatef in front of it: the engine composes the modelview like the GL matrix stack does. */
typedef struct {
    float v[4][3];              /* glVertex3f   */
    float uv[4][2];             /* glTexCoord2f */
    float n[3];                 /* glNormal3f   */
    int32_t nv;                 /* 3 or 4, | MW_POLY_* flags */
    int32_t tex;                /* texture id from mw_upload_texture, -1 = untextured */
    float rgb[3];               /* glColor3f    */
    float xf[4];                /* MW_POLY_XF: translation x, y, z and rotation angle in degrees about +y (entity.py:205-207) */
} mw_poly;

/* Host view of the world state of `count` consecutive envs; any pointer may be NULL
 * (= leave / do not fetch).  Mirrors the Python attributes the reference keeps:
 * agent.pos/dir/cam_* (entity.py:455-515), env.sky_color/light_* (miniworld.py:576-578),
 * entities[*].pos/dir/size/color_vec/scale/radius/height (entity.py), step_count, carrying. */
typedef struct {
    double *agent_pos;          /* [count][3]                                        */
    double *agent_dir;          /* [count]                                           */
    double *cam;                /* [count][4] cam_height, cam_fwd_disp, cam_pitch(deg), cam_fov_y(deg) */
    double *light;              /* [count][12] sky_color, light_pos, light_color, light_ambient */
    int32_t *carrying;          /* [count] entity slot or -1                         */
    int32_t *step_count;        /* [count]                                           */
    int32_t *num_picked_up;     /* [count]                                           */
    int32_t *ent_kind;          /* [count][max_ents] MW_ENT_* (NONE = empty / removed) */
    int32_t *ent_mesh;          /* [count][max_ents] mesh id                         */
    int32_t *ent_static;        /* [count][max_ents]                                 */
    double *ent_pos;            /* [count][max_ents][3]                              */
    double *ent_dir;            /* [count][max_ents]                                 */
    double *ent_geom;           /* [count][max_ents][9] size xyz, color rgb, scale, radius, height */
    double *extent;             /* [count][4] env.min_x, max_x, min_z, max_z (miniworld.py:588-591); top view only */
} mw_state_view;

/* ---- placement programs (MW_GEN_PROGRAM) --------------------------------------------------------------------
 * The env families beyond the four BASELINE configs have a FIXED floorplan; their _gen_world only draws a few
 * numbers and places entities (fourrooms.py:46-73, tmaze.py:54-81, ymaze.py:56-108, wallgap.py:48-77, threerooms.py:47-73,
 * putnext.py:45-65, roomobjects.py:44-80, sidewalk.py:51-91, sign.py:101-150).  The host compiles that method into
 * the table below; the device generator executes it on the env's random stream (MW_RNG_PCG64: numpy's own, so env i
 * is the reference's reset(seed + i)), for mw_reset and for the same-step auto-reset:
 *   1. the entity table is initialised from the template (ent_*: what the constructors give before placement);
 *   2. the ops run in order; the first placement op also runs Room._gen_static_data for every room (miniworld.py:856-857):
 *      with domain_rand three texture-variant draws per room, wall / floor / ceiling (opengl.py:134-138), after which
 *      the room polygons of the template are re-emitted into the env's own geometry set with the variants' texture ids
 *      and texture coordinates (metres * TEX_DENSITY / size, miniworld.py:82-119);
 *   3. the per-episode parameters, Box.randomize for every box in slot order, Agent.randomize (miniworld.py:576-585). */
#define MW_PROG_MAX_ROOMS 16
#define MW_PROG_MAX_TEX 8
#define MW_PROG_MAX_OPS 48
#define MW_PROG_MAX_ENTS 64

enum {
    MW_OP_COIN = 1,         /* reg = np_random.integers(0, n)              n in `slot`                          */
    MW_OP_DRAW_DIR = 2,     /* dir_reg = np_random.uniform(-dir, dir)      (an argument evaluated before place_entity) */
    MW_OP_PLACE = 3,        /* place_entity(ent, room=..., min_x=... ) by rejection sampling (miniworld.py:839-909)  */
    MW_OP_FIXED = 4,        /* place_entity(ent, pos=(lx, a, lz), dir=...)                                       */
    MW_OP_BOX_SIZE = 5,     /* Box(size=np_random.uniform(a, b)): size, radius, height of box `slot`             */
    MW_OP_COLOR = 6,        /* colour index = np_random.choice(6) for `slot`: room = 0 box (colour vector),
                             * 1 / 2 ball / key (mesh id = flags + index)                                        */
    MW_OP_APPEND = 7        /* self.entities.append(ent): in the list from here on (no draw, no static data)     */
};

typedef struct {
    int32_t nverts;             /* 3 or 4 outline corners, counter-clockwise seen from above (miniworld.py:127-176) */
    int32_t wall_tex, floor_tex, ceil_tex;      /* indices into mw_gen_program.tex_* (texture NAMES)             */
    double ox[4], oz[4];        /* Room.outline                                                                   */
    double nx[4], nz[4];        /* Room.edge_norms (for point_inside, miniworld.py:272-284)                       */
    double min_x, max_x, min_z, max_z;
    double cdf;                 /* cumulative room probability: np_random.choice(len(rooms), p=room_probs) picks
                                 * searchsorted(cdf, u, side="right") (miniworld.py:873-875)                      */
} mw_prog_room;

typedef struct {
    int32_t op;                 /* MW_OP_*                                                                         */
    int32_t slot;               /* entity slot; -1 = the agent                                                     */
    int32_t room;               /* PLACE: room index, -1 = choice over all rooms                                   */
    int32_t cond;               /* run only if the coin register == cond (-1: always)                              */
    int32_t dir_mode;           /* 0: uniform(-pi, pi) drawn after the position; 1: `dir`; 2: the DRAW_DIR register */
    int32_t flags;              /* PLACE: bit 0..3 = min_x, max_x, min_z, max_z given (lx, hx, lz, hz)             */
    double lx, hx, lz, hz;
    double dir;
    double a, b;
} mw_prog_op;

typedef struct {
    int32_t n_rooms, n_tex, n_ops, n_ents;
    mw_prog_room rooms[MW_PROG_MAX_ROOMS];
    int32_t tex_nvar[MW_PROG_MAX_TEX];          /* variants of each texture name (Texture.get, opengl.py:124-140) */
    int32_t tex_var_id[MW_PROG_MAX_TEX][9];     /* their texture ids (mw_upload_texture)                           */
    double tex_var_scale[MW_PROG_MAX_TEX][9][2];/* TEX_DENSITY / (width, height)                                   */
    mw_prog_op ops[MW_PROG_MAX_OPS];
    int32_t ent_kind[MW_PROG_MAX_ENTS], ent_mesh[MW_PROG_MAX_ENTS], ent_static[MW_PROG_MAX_ENTS];
    double ent_pos[MW_PROG_MAX_ENTS][3], ent_dir[MW_PROG_MAX_ENTS], ent_geom[MW_PROG_MAX_ENTS][9];
    double colors[6][3];        /* COLORS of the six sorted colour names (entity.py:30-43), for MW_OP_COLOR         */
    double extent[4];           /* env.min_x, max_x, min_z, max_z                                                  */
    double street[4];           /* MW_TASK_SIDEWALK: min_x, max_x, min_z, max_z of the forbidden room              */
    int32_t sign_n, pad;        /* MW_TASK_SIGN: objects in the order sign.py:160-169 visits them                  */
    int32_t sign_slot[8];
    double sign_reward[8];
} mw_gen_program;

/* Installs the placement program of an engine created with MW_GEN_PROGRAM.  The template geometry (what
 * mw_set_geometry would receive for domain_rand = 0) comes with, per polygon, the room it belongs to (-1: not a room
 * polygon, copied as it is), its surface (0 wall, 1 floor, 2 ceiling) and the metre coordinates its texture
 * coordinates are computed from (poly_m[p][k] = the two factors of vertex k), so that a texture-variant draw can
 * re-emit it.  With shared_geometry = 1 (no texture randomisation) the template is installed as the shared set. */
int mw_set_gen_program(mw_engine *e, const mw_gen_program *prog, const mw_poly *polys, const int32_t *poly_room,
                       const int32_t *poly_surf, const double *poly_m /* [n_polys][4][2] */, int32_t n_polys,
                       const double *segs, int32_t n_segs);

/* ---- lifetime --------------------------------------------------------------- */
/* replaces MiniWorldEnv.__init__'s GL setup (shadow window, FrameBuffer) miniworld.py:504-518 */
int mw_create(const mw_config *cfg, mw_engine **out);
void mw_destroy(mw_engine *e);
/* last error message of `e` (or of the failed mw_create when e == NULL) */
const char *mw_last_error(const mw_engine *e);

/* ---- assets ----------------------------------------------------------------- */
/* replaces Texture.load (opengl.py:148-184): RGB8, rows bottom-up; builds the mip pyramid */
int mw_upload_texture(mw_engine *e, int32_t tex_id, const uint8_t *rgb_bottom_up, int32_t w, int32_t h);
/* replaces ObjMesh's vertex lists (objmesh.py:139-207): per-face-vertex arrays [ntris][3][k] (pos, nrm, rgb:
 * k = 3; uv: k = 2); tex_id = the chunk's map_Kd texture (objmesh.py:209-216, 280-292) or -1 */
int mw_upload_mesh(mw_engine *e, int32_t mesh_id, const float *pos, const float *nrm, const float *uv,
                   const float *rgb, int32_t ntris, int32_t tex_id);

/* ---- world ------------------------------------------------------------------ */
/* replaces Room._gen_static_data + _render_static's display list (miniworld.py:286-399,
 * 1019-1062) and env.wall_segs (:998-999).  env = -1 for the shared set.
 * segs: [n_segs][2][2] (x,z of both endpoints). */
int mw_set_geometry(mw_engine *e, int32_t env, const mw_poly *polys, int32_t n_polys,
                    const double *segs, int32_t n_segs);
/* state injection / inspection (synchronous) */
/* reads one geometry set back (polys: max_polys entries, segs: max_segs*4 doubles) */
int mw_get_geometry(mw_engine *e, int32_t env, mw_poly *polys, int32_t *n_polys, double *segs, int32_t *n_segs);
int mw_set_state(mw_engine *e, int32_t first_env, int32_t count, const mw_state_view *host);
/* Test hook: host double[num_envs][3] = forward_step, forward_drift, turn_step to use in
 * the next steps instead of the defaults / device RNG draws (miniworld.py:678-680);
 * NULL switches the override off. */
int mw_set_step_params(mw_engine *e, const double *host_params);
int mw_get_state(mw_engine *e, int32_t first_env, int32_t count, mw_state_view *host);
/* device-side MiniWorldEnv.reset (miniworld.py:544-604) for the configured generator.
 * mask: host uint8[num_envs] or NULL (= all); seeds: host uint64[num_envs] or NULL.
 * With MW_GEN_NONE (host-generated worlds) only the re-seeding happens: seeds[i] (masked) re-seeds env i's device
 * stream, which serves the per-step domain-randomisation draws (miniworld.py:677-680); seeds == NULL is an error. */
int mw_reset(mw_engine *e, const uint8_t *mask, const uint64_t *seeds, void *stream);

/* ---- the hot path ------------------------------------------------------------ */
/* MiniWorldEnv.step (miniworld.py:670-730) + env rule + render_obs (:1177-1221) for all envs.
 *   d_actions int32[N]            MiniWorldEnv.Actions (:451-468)
 *   d_obs     uint8[N][H][W][3]   FrameBuffer.resolve() layout, row 0 = top (opengl.py:339-398)
 *   d_depth   float[N][H][W][1]   FrameBuffer.get_depth_map(0.04, 100) (opengl.py:400-435); NULL = skip
 *   d_reward  float[N], d_term uint8[N], d_trunc uint8[N]
 * All device pointers; asynchronous on `stream`. */
int mw_step(mw_engine *e, const int32_t *d_actions, uint8_t *d_obs, float *d_depth,
            float *d_reward, uint8_t *d_term, uint8_t *d_trunc, void *stream);
/* Layout of the d_obs buffer written by mw_step / mw_render / mw_render_top — the reference's
 * observation wrappers (wrappers.py) folded into the raster kernel's store:
 *   MW_OBS_HWC_U8   uint8 [N][H][W][3]   the env's own observation (default)
 *   MW_OBS_CWH_U8   uint8 [N][3][W][H]   PyTorchObsWrapper.observation: transpose(2, 1, 0)   (wrappers.py:11-25)
 *   MW_OBS_GREY_F64 double[N][H][W][1]   GreyscaleWrapper.observation: 0.30 R + 0.59 G + 0.11 B evaluated as
 *                                        numpy does, in float64, left to right             (wrappers.py:28-46) */
enum { MW_OBS_HWC_U8 = 0, MW_OBS_CWH_U8 = 1, MW_OBS_GREY_F64 = 2 };
int mw_set_obs_layout(mw_engine *e, int32_t layout);

/* Test hook, host only (no device, no engine): the first n draws of the MW_RNG_PCG64 stream for `seed`, with the very
 * functions the device generators inline.  bounds[i] == 0 (or bounds == NULL): a double in [0, 1), i.e.
 * Generator(PCG64(SeedSequence(seed))).random(); bounds[i] = k > 0: an integer in [0, k), i.e. Generator.integers(0, k)
 * / Generator.choice(k), returned as a double. */
int mw_pcg64_draws(uint64_t seed, int32_t n, const int32_t *bounds, double *out);

/* Test hooks, GPU only, no engine (tests/test_gpu_numerics.py): the kernels' 3-instruction reciprocal / quotient
 * (hardware estimate + fused Newton / Markstein steps, mw_raster_common.h) against the IEEE division the oracle
 * performs — mw_selftest_rcp over all 2^32 floats (bad_per_exp[512]: mismatches per sign|exponent, examples[64],
 * *n = their total), mw_selftest_div over 2^32 pseudo-random pairs of its domain. */
int mw_selftest_rcp(unsigned long long *bad_per_exp, uint32_t *examples, uint32_t *n);
int mw_selftest_div(unsigned long long *n_bad, uint32_t *examples);
/* ... and the geometry kernel's visiting-order sort (mw_geom.hip: up to 512 keys per block, bitonic, in registers) on the
 * caller's keys: keys[blocks][512], n[blocks], order[blocks][513] (order[b][1 + k] = low 16 bits of block b's k-th smallest key). */
int mw_selftest_sort(const uint32_t *keys, const int32_t *n, int32_t blocks, uint16_t *order);
/* ... and two shortcuts of the quad raster kernel (mw_rasterq.hip) for all 2^32 floats: n_bad[0] = inputs where
 * v_cvt_pk_u8_f32(x * (255 / S)) differs from FrameBuffer.resolve()'s unorm8 conversion of x * (1 / S) (S = 4, 8);
 * n_bad[1] = inputs where the lod taken from rho^2's bits differs from llvmpipe's float arithmetic (pyramids of 1 .. 12
 * levels); examples[2][32] = the first offending bit patterns of each. */
int mw_selftest_q(unsigned long long *n_bad, uint32_t *examples);

/* render_obs / render_depth only (miniworld.py:1177-1236) */
int mw_render(mw_engine *e, uint8_t *d_obs, float *d_depth, void *stream);
/* render_top_view (miniworld.py:1088-1175): orthographic map of the whole floorplan into the same
 * kind of buffers; render_agent != 0 also draws Agent.render's marker (entity.py:518-539) */
int mw_render_top(mw_engine *e, uint8_t *d_obs, float *d_depth, int32_t render_agent, void *stream);
/* render() / render_obs(vis_fb) / render_top_view(vis_fb) (miniworld.py:1340-1362): ONE env into a frame
 * buffer of any size (multiples of 16 x 4) with msaa = 1, 4, 8 or 16 samples (vis_fb = FrameBuffer(800, 600, 16),
 * miniworld.py:518).  view_flags: bit 0 top view, bit 1 draw the agent marker.
 *   d_out uint8[height][width][3], d_depth float[height][width] or NULL.  Not the hot path. */
int mw_render_view(mw_engine *e, int32_t env, int32_t view_flags, int32_t width, int32_t height, int32_t msaa,
                   uint8_t *d_out, float *d_depth, void *stream);
/* get_visible_ents (miniworld.py:1238-1333) for envs [first_env, first_env + count): rooms drawn depth-only
 * into the obs frame, then one GL_ANY_SAMPLES_PASSED query per entity around its 0.2 m proxy box, in
 * entity-slot (= self.entities) order with depth writes on.  d_vis uint8[count][max_ents], 1 = visible;
 * empty slots report 0.  Needs obs_width * obs_height * 32 bytes of LDS (<= 160 KiB).  Not the hot path. */
int mw_visible_ents(mw_engine *e, int32_t first_env, int32_t count, uint8_t *d_vis, void *stream);

/* checks the device-side status word (capacity overflows); synchronises `stream` */
int mw_check(mw_engine *e, void *stream);

/* ---- measurement ------------------------------------------------------------- */
/* average duration (ms) of the dominant (raster) kernel and of the setup kernel over the launches since
 * the last call, measured with HIP events on the stream the kernels ran on; enables timing on first use.
 * reset > 0: from now on one launch in `reset` is bracketed with events (1 = every launch; recording on every
 * launch costs a few percent of the step rate); reset = 0: the default, one in 8; reset < 0 switches timing off.
 * `launches` is the number of launches measured.  Returns <0 on error. */
int mw_kernel_time_ms(mw_engine *e, int32_t reset, double *raster_ms, double *setup_ms, int64_t *launches);

/* Which raster kernels drew the last frame (the FrameBuffer half of render_obs, opengl.py:202-398) — so that a test can say
 * which code its fixtures exercised: MW_PATH_QUAD the quad kernel (mw_rasterq.hip: small scenes, 8 or 4 samples),
 * MW_PATH_QUAD_MESH the same for every tile no mesh entity can touch + the mesh-aware tile kernel for the others,
 * MW_PATH_TILE the tile kernels (mw_raster.hip: big scenes, MW_K2Q=0), MW_PATH_GENERIC the generic-resolution kernels
 * (other sample counts, frames beyond 128 x 96 pixels (W H > 12 288: the tile kernels' 32-bit edge sums), MW_GENERIC_RASTER=1); -1 before the first frame. */
/* The `info` dict of the envs' step() as device arrays, asynchronous on `stream` (either pointer may be NULL):
 *   d_health  int32[N]     CollectHealth: info["health"] (collecthealth.py:100)
 *   d_ent_pos double[N][3] position of entity slot `ent_slot`: TMaze / YMaze info["goal_pos"] = box.pos (tmaze.py:89, ymaze.py:125)
 * Values are those of the state the device holds: with MW_AUTORESET_SAME_STEP an env that just finished reports its new episode.
 * d_health on an engine whose task is not MW_TASK_COLLECT is MW_E_INVALID (there is no health array). */
int mw_get_info(mw_engine *e, int32_t *d_health, double *d_ent_pos, int32_t ent_slot, void *stream);
/* The same two values as they stood when each env's LAST FINISHED episode ended (collecthealth.py:100: the health that ended it;
 * tmaze.py:89 / ymaze.py:125: that episode's goal_pos = position of entity slot mw_config.goal_ent) — with MW_AUTORESET_SAME_STEP the step
 * kernel keeps them before it installs the next world (Gymnasium's `final_info` of a same-step vector env).  Undefined for an env that
 * has not finished an episode yet; either pointer may be NULL; d_health needs MW_TASK_COLLECT. */
int mw_get_final_info(mw_engine *e, int32_t *d_health, double *d_goal_pos, void *stream);

/* Diagnostic (synchronises `stream`): how many triangles the last frame's display list held per env after clipping and culling —
 * what max_visible has to pay for (6 records per unit), and what decides which raster kernel an env's frame takes.
 * The stored length is clamped to the list's capacity (6 x max_visible): a value EQUAL to the capacity means "at least this
 * many" — raise max_visible and look again (mw_check reports the overflow itself as MW_E_CAPACITY). */
int mw_get_list_lengths(mw_engine *e, int32_t first_env, int32_t count, int32_t *host_out, void *stream);

/* Test hook: the sequence number of the next frame with mesh entities.  Its low 16 bits stamp the per-pixel chains of the
 * fragments of mesh triangles that cross a frustum plane (a head of another stamp reads as empty; the heads are wiped on the
 * frame whose stamp is 0); a test moves it to the wrap instead of rendering 65 536 frames.  The parity must stay (the frame's
 * work lists alternate with it). */
int mw_debug_set_mesh_frame_seq(mw_engine *e, uint32_t seq);
/* ... and the chain heads themselves, uint32[N][H][W] = stamp << 16 | newest fragment of the pixel + 1 (synchronises `stream`). */
int mw_debug_get_slow_heads(mw_engine *e, uint32_t *host_out, void *stream);

enum { MW_PATH_TILE = 0, MW_PATH_QUAD = 1, MW_PATH_QUAD_MESH = 2, MW_PATH_GENERIC = 3 };
int mw_raster_path(const mw_engine *e);

#ifdef __cplusplus
}
#endif
#endif
