// mwengine internal types shared by the host runtime (mw_engine.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mwengine.h"

#define MW_MAX_TEX 64
#define MW_MAX_MESH 32
#define MW_MAX_LEVELS 16
#define MW_RASTER_REC 64     // dwords per raster record
#define MW_SHADE_REC 32      // dwords per shade record (attribute planes, colour, tex, depth plane)
#define MW_CULL_REC 24       // a[4] b[4] c[4] tmin[4] tmax[4] flags pad[3]
#define MW_LDS_RECS 32       // triangle records a small-scene raster wave stages in LDS ...
#define MW_LDS_SHADE_Q 7      // ... as the quads K2 reads of each: 7 of the shade record's 8,
#define MW_LDS_CULL_Q 5       //     5 of the classification record's 6 (192 B per triangle: 7.5 KB + 192 B per wave, 5 waves per SIMD)
#define MW_TILE_W 16
#define MW_TILE_H 4
#define MW_SKY_PID 0xFFFFu
#define MW_ENVHDR 640         // floats per env: sky, light colours, mesh-entity table (geometry kernel -> raster kernels)
#define MW_MAX_MESH_ENTS 21   // mesh entities drawn per env
#define MW_HDR_MESH 32        // first float of the mesh-entity table
#define MW_K1_PROF_SLOTS 32
// mesh path (mw_mesh.h, mw_raster_mesh.hip)
#define MW_PLANE_REC 16         // plane cache record, one 64-byte sector: (w plane, tex) (r plane, state) (g plane, s.a0) (b plane, s.dadx);
#define MW_PLANE_XTRA 4         //   a textured mesh's fifth quad (s.dady, t plane) lives in a second array behind the records ([N][cap][4]):
                                //   at a stride of 80 bytes a record straddled two sectors three times in four (2 x the write traffic)
#define MW_PIECE_REC 20         // a slow-path piece's record: the five quads in a row
#define MW_PLANE_SLOW 2         // state: the triangle crosses a frustum plane — its fragments come from the env's slow-fragment list
#define MW_SLOW_TRIS 1024       // per env: mesh triangles that cross a frustum plane (a mesh at the frame's edge)
#define MW_SLOW_FRAGS 8191      // per env: their fragments, (draw id << 16 | piece of the fan << 13 | next fragment of the pixel + 1, piece's record), chained per pixel
#define MW_SLOW_PIECES (MW_SLOW_TRIS * 7)       // per env: the attribute planes of the pieces of their fans (plane-cache records), 7 places per listed triangle
#define MW_SLOW_STRIDE (MW_SLOW_FRAGS + 1 + MW_SLOW_PIECES * (MW_PIECE_REC / 4))       // float4s per env: fragments, then pieces
#define MW_OCC_CACHE_HDR 8
// floats per set: header, 8 per wall, 8 per box of eight polygons; whole 128-byte lines
#define MW_OCC_CACHE_STRIDE(max_polys) ((MW_OCC_CACHE_HDR + 8 * (size_t)(max_polys) + 8 * (size_t)(((max_polys) + 7) / 8) + 31) / 32 * 32)
#define MW_HDR_MESH_STRIDE 28 // floats per entry: slot, first draw id, triangles, first triangle, texture, normal scale, light[3], mvp[16], mesh triangles drawn before, tile rectangle, mesh id
#define MW_MESH_VCAP 3568       // distinct positions of a mesh whose vertex stage runs per vertex (mw_mesh_entity_kernel: 16 bytes of LDS each, 57 088 B + the kernel's 8 204 B of queues = 65 292 B, within a workgroup's 64 KB; static_assert in mw_raster_mesh.hip)
#ifndef MW_ENT_THREADS
#define MW_ENT_THREADS 512      // lanes of the mesh entity kernel's workgroup
#endif
#define MW_ENT_ROUND 2048       // triangles between two drains of its winner queue
#define MW_ENT_VPL ((MW_MESH_VCAP + MW_ENT_THREADS - 1) / MW_ENT_THREADS)     // positions per lane of the vertex stage
#define MW_ENT_TPL (MW_ENT_ROUND / MW_ENT_THREADS)      // triangles per lane and round
#ifndef MW_ENT_OCC
#define MW_ENT_OCC 6            // wavefronts per SIMD the entity kernel is compiled for: 80 registers (17 spilled: ~30 MB of scratch traffic per step) — at 96 or 105 without spills a quad-kernel workgroup no longer fits beside two of its workgroups on a CU and the frame loses 25 us
#endif
#define MW_CNT_LONG 0           // counters of the mesh path's work lists (MwArgs::ent_list_n): per XCD long / short meshes in view and the
#define MW_CNT_SHORT 8          //   entity kernel's cursor into them, the mesh tiles, the envs with slow-path triangles
#define MW_CNT_CURSOR 16
#define MW_CNT_TILES 32         // (8: per XCD, like the entities)
#define MW_CNT_SLOW_ENVS 40
#define MW_CNT_WORDS 64
#define MW_ENT_BIG_PIXELS 48      // a triangle whose bounding box holds more pixels is rasterised by a wavefront, a pixel per lane, instead of by one lane

// status bits written by kernels, read by mw_check()
#define MW_ST_VIS_OVERFLOW 1u
#define MW_ST_PLACEMENT_FAIL 2u

struct MwTexDesc {
    uint32_t w, h, nlevels, pad;
    // per mip level, everything a bilinear fetch needs, so that a lane gets it with two 16-byte loads instead of
    // shifting / clamping / converting the level-0 size itself (8 VALU instructions per level and fetch)
    struct Level {
        uint32_t off;              // first texel (dword index into the texel pool)
        uint32_t w;                // row length in texels
        uint32_t wmask, hmask;     // w - 1, h - 1 (wrap masks of power-of-two levels)
        float fw, fh;              // (float)w, (float)h
        uint32_t h, pad;
    } lvl[MW_MAX_LEVELS];
};

struct MwMeshDesc {
    uint32_t ntris;
    int32_t tex;
    uint32_t first;                // first triangle in the mesh pools
    uint32_t bound_bits;           // float bits: max |vertex| (radius of the bounding sphere about the mesh origin)
    float last_n[3];               // vertex normal of the LAST triangle's last vertex in drawing order (GL's current
    uint32_t pad;                  //   normal after the mesh, for the top view's agent marker)
    uint32_t vfirst, nverts;       // the mesh's table of distinct positions in the vertex pool (nverts = 0: more than MW_MESH_VCAP, no table)
    float bmin[3], bmax[3];        // bounding box of the vertices (object space), its centre and the radius of the sphere about the centre
    float center[3];               //   that holds them: the geometry kernel's view test and tile rectangle (a ball's origin lies at its
    float radius;                  //   foot: the sphere about the ORIGIN has twice the ball's radius, four times its tiles)
};

// Mesh pools: per-face-vertex arrays in drawing order (= draw ids, GL's first-drawn-wins on equal depth, the oracle's
// triangle indices).  The mesh kernels RASTERISE the triangles in another order — sorted by the direction of their face
// normal (mw_upload_mesh), so that the 64 triangles of a wavefront face the same way and back-face culling retires whole
// waves instead of half the lanes of each: a position record is 9 floats + one word, the index of the i-th triangle of
// that order.  (Shading looks a triangle up by its draw id directly: no indirection on the tile phase's critical path.)
#define MW_MESH_POS_STRIDE 10

// Generator tables; kept in device memory because dynamic indexing into a by-value kernarg
// struct would force a private (scratch) copy of the whole argument block.
struct MwGenTables {
    double gen_tab[12];     // PICKUP: per kind (ball, box, key): radius, height, scale, first mesh id; MAZE: see mw_gen.h
    double gen_colors[18];  // PICKUP: COLORS of the 6 sorted colour names (entity.py:30-40); MAZE: texcoord scales
    int32_t tex_nvar[3];    // texture domain randomisation of generated rooms: wall, floor, ceiling
    int32_t tex_var_id[3][9];
    double tex_var_scale[3][9][2];
    double room_wall_height;
    int32_t room_no_ceiling, pad2;
};

// The pre-generated next world of every env ("spare"): same layouts as the live arrays.  With a device generator
// and no domain randomisation an episode's end only copies the spare into place; extra blocks of the NEXT step's K1
// regenerate it while the regular blocks step (generate_world with the state pointers redirected here).  The random stream
// is consumed in the same order — worlds only — so seed-exactness is unaffected; with domain randomisation the
// per-step draws interleave with the worlds in the stream, and the generator stays inline.
struct MwSpare {
    double *ax, *ay, *az, *adir, *cam, *light, *extent;
    int32_t *ekind, *emesh, *estatic;
    double *epos, *edir, *egeom;
    mw_poly *polys; int32_t *npolys; double *segs; int32_t *nsegs;     // per-env geometry sets only
};

// Device side of a placement program (mw_set_gen_program): the table itself plus the template geometry.
struct MwProgram {
    mw_gen_program p;
    const mw_poly *polys;       // [n_polys] template polygons
    const int32_t *poly_room;   // [n_polys]
    const int32_t *poly_surf;   // [n_polys]
    const double *poly_m;       // [n_polys][4][2]
    const double *segs;         // [n_segs][4]
    int32_t n_polys, n_segs;
};

// Everything the kernels need; passed by value (kernarg).
struct MwArgs {
    int32_t N, W, H, E;
    int32_t max_polys, max_segs, max_vis, shared_geom;
    int32_t task, goal_ent, num_objs, max_steps;
    int32_t domain_rand, generator, autoreset, tiles_x;
    int32_t tiles_y, n_tiles, goal_ent2, env_base;    // env_base: first env of this launch (0 for the batched step)
    int32_t rng_mode, occlusion;    // occlusion: the geometry kernel of big scenes drops room polygons hidden behind full-height walls (mw_geom.hip; MW_OCCLUSION=0 turns it off)
    double agent_radius, max_forward_step, agent_height;
    mw_range fwd, drift, turn;
    mw_range sky[3], light_pos[3], light_color[3], light_ambient[3], color_bias[3];
    mw_range cam_height, cam_fwd_disp, cam_pitch, cam_fov_y;
    double gen_args[8];
    const MwGenTables *gt;  // generator tables (device memory: they are indexed dynamically)
    const MwProgram *prog;  // MW_GEN_PROGRAM / MW_TASK_SIDEWALK / MW_TASK_SIGN tables, else null
    // --- world state, SoA over envs -------------------------------------------------
    double *ax, *ay, *az, *adir;
    double *cam;        // [4][N]
    double *light;      // [12][N]
    int32_t *carry, *step, *picked;
    int32_t *health;    // [N] MW_TASK_COLLECT (collecthealth.py:77, 83)
    // what `info` held when an env's last episode ended — the same-step auto-reset installs the next world in the same kernel, so
    // K1 keeps the finished episode's values here: health (MW_TASK_COLLECT) and the position of entity slot goal_ent (TMaze / YMaze:
    // info["goal_pos"])
    int32_t *final_health;  // [N]
    double *final_goal;     // [3][N]
    int32_t *ekind, *emesh, *estatic;   // [E][N]
    double *epos;       // [3][E][N]
    double *edir;       // [E][N]
    double *egeom;      // [9][E][N]
    double *extent;     // [4][N] world extents min_x, max_x, min_z, max_z (top view)
    uint64_t *rng;      // [5][N]  Philox: seed, counter; PCG64: state hi, lo, increment hi, lo, buffered uint32
    const double *step_override;        // [N][3] or null
    // --- geometry -------------------------------------------------------------------
    const mw_poly *polys;   // [sets][max_polys]
    const int32_t *npolys;  // [sets]
    const double *segs;     // [sets][max_segs][4]
    const int32_t *nsegs;   // [sets]
    // --- assets ---------------------------------------------------------------------
    const MwTexDesc *tex;
    const uint32_t *texels; // RGBA8 pool
    const MwMeshDesc *mesh;
    const float *mesh_pos;  // [tris][3][3]
    const float *mesh_nrm;  // [tris][3][3]
    const float *mesh_rgb;  // [tris][3][3]
    const float *mesh_uv;   // [tris][3][2]
    // --- per-step scratch -----------------------------------------------------------
    float *rec_raster;      // [N][max_vis][64]
    float *rec_shade;       // [N][max_vis][16]
    float *rec_cull;        // [N][max_vis][MW_CULL_REC] per-primitive tile classification data
    int32_t *nvis;          // [N]
    uint16_t *rec_order;    // [N][max_vis + 1] big scenes only: [0] sorted flag, then list indices by ascending depth bound
    float *envhdr;          // [N][MW_ENVHDR]
    uint32_t *status;
    const MwSpare *spare;   // device copy of the spare pointers, null = generator inline
    uint32_t *refill_mask;  // [N] 0 spare ready, 1 consumed (refill pending), 2 refill running, 3 env regenerating inline
    const MwArgs *gen_live; // device copies of this struct for the generators (live state / spare state): they index
    const MwArgs *gen_spare;//   it dynamically, which a by-value kernarg would turn into a scratch copy
    int32_t *pending_remove; // [N] entity slot that leaves the list after this step's frame (-1 none): written by K1, applied by the geometry kernel
    // big scenes: what the geometry kernel's culling derives from a world's polygons alone (mw_geom.hip), kept from frame to
    // frame.  occ_valid[set]: polygon count + 1 of the world the cache belongs to, 0 after anything rewrote the polygons.
    int32_t *occ_valid;     // [sets] or null
    float *occ_cache;       // [sets][MW_OCC_CACHE_STRIDE(max_polys)]
    // the frame's mesh entities in view, for mw_mesh_entity_kernel (written by the geometry kernel when non-null): one pair of
    // lists per XCD — env e's entities go to the lists of XCD e % n_xcc, and the entity kernel's workgroups draw from the lists of
    // the XCD they run on, so that an env's sample keys are touched from ONE XCD and their atomic minima can stay in its L2 —:
    // ent_list_n[MW_CNT_LONG + x] entries env | table entry << 24 at ent_list + x * ent_list_cap (meshes of 1024 triangles and more:
    // drawn from first), ent_list_n[MW_CNT_SHORT + x] at ent_list + (8 + x) * ent_list_cap (the others).
    // ... and the tiles inside the union of their tile rectangles, for the mesh tiles' launch: ent_list_n[MW_CNT_TILES + x] entries
    // env | tile << 24 at tile_list + x * tile_list_cap (env e under x = e % n_xcc: the wavefronts b % n_xcc == x of the launch
    // draw them, so that an env's records and keys are fetched into one XCD's L2).  (ent_list_n: MW_CNT_WORDS counters per frame parity.)
    uint32_t *ent_list;
    int32_t *ent_list_n;
    uint32_t *tile_list;
    int32_t ent_list_cap, tile_list_cap;
    int32_t n_xcc, pad_xcc;     // XCDs of the device as the engine's probe found them (1, 2, 4 or 8), mw_xcc_probe_kernel
    unsigned long long *k1_prof;   // MW_K1_PROF: [N][MW_K1_PROF_SLOTS] cycle counters of the geometry kernel's phases, start and end time of the env's wavefront (tools/perf/kgprof.py; perf experiments only), else null
};
