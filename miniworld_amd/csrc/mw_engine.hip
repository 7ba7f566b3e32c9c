// mwengine host runtime: the C ABI of include/mwengine.h on top of the HIP kernels.
// Owns the device-resident Structure-of-Arrays world state of N environments, the texture /
// mesh pools and the per-step scratch; never touches torch (the caller hands raw device
// pointers and a hipStream_t).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <map>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mw_device.h"
#include "mw_rng.h"

extern "C" __global__ void mw_step_setup_kernel(MwArgs a, int do_step, int view_flags, const int32_t *actions,
                                                float *reward, uint8_t *term, uint8_t *trunc);
extern "C" __global__ void mw_step_setup_pcg_kernel(MwArgs a, int do_step, int view_flags, const int32_t *actions,
                                                    float *reward, uint8_t *term, uint8_t *trunc);
extern "C" __global__ void mw_reset_pcg_kernel(MwArgs a, const uint8_t *mask, int force_all, int mark_refill);
extern "C" __global__ void mw_step_setup_dense_kernel(MwArgs a, int do_step, int lanes_per_env, const int32_t *actions,
                                                       float *reward, uint8_t *term, uint8_t *trunc);
extern "C" __global__ void mw_step_setup_dense_pcg_kernel(MwArgs a, int do_step, int lanes_per_env, const int32_t *actions,
                                                           float *reward, uint8_t *term, uint8_t *trunc);
extern "C" __global__ void mw_geom_kernel(MwArgs a, int view_flags, int S, int L, int n_env);
extern "C" __global__ void mw_geom_any_kernel(MwArgs a, int view_flags, int S, int L, int n_env);
extern "C" __global__ void mw_geom_big_kernel(MwArgs a, int view_flags, int S, int L, int n_env);
extern "C" __global__ void mw_geom_big_any_kernel(MwArgs a, int view_flags, int S, int L, int n_env);
#define MW_RASTER_DECL(name) \
    extern "C" __global__ void name(int N, int W, int H, int max_vis, int tiles_x, int n_tiles, int waves_per_env, int tiles_per_wave, \
                                    const float *rec_raster, const float *rec_shade, const float *rec_cull, const int32_t *nvis, \
                                    const float *envhdr, const MwTexDesc *texd, const uint32_t *texels, uint8_t *obs, float *depth, int dbg, \
                                    int texel_bytes, const uint16_t *rec_order, const float *mesh_pos, const float *mesh_nrm, \
                                    const float *mesh_rgb, const float *mesh_uv, uint32_t *mesh_keys, const float *plane_cache, int plane_cap, \
                                    const float4 *slow_frags, const uint32_t *slow_head, const uint32_t *tile_list, int32_t *tile_n, int tile_list_cap, int n_xcc)
MW_RASTER_DECL(mw_raster_kernel);
MW_RASTER_DECL(mw_raster_depth_kernel);
MW_RASTER_DECL(mw_raster_big_kernel);
MW_RASTER_DECL(mw_raster_big_depth_kernel);
MW_RASTER_DECL(mw_raster_wrap_kernel);
MW_RASTER_DECL(mw_raster_big_wrap_kernel);
MW_RASTER_DECL(mw_raster_mesh_kernel);
MW_RASTER_DECL(mw_raster_nomesh_kernel);
MW_RASTER_DECL(mw_raster_nomesh_depth_kernel);
MW_RASTER_DECL(mw_raster_mesh_depth_kernel);
MW_RASTER_DECL(mw_raster_mesh_wrap_kernel);
MW_RASTER_DECL(mw_raster_big_mesh_wrap_kernel);
#define MW_RASTERQ_DECL(name) \
    extern "C" __global__ void name(int N, int W, int H, int max_vis, int tiles_x, int n_tiles, const float *rec_raster, const float *rec_shade, \
                                    const float *rec_cull, const int32_t *nvis, const float *envhdr, const uint32_t *texels, uint8_t *obs, \
                                    float *depth, int dbg, int texel_bytes, unsigned long long *prof)
MW_RASTERQ_DECL(mw_rasterq_kernel);
MW_RASTERQ_DECL(mw_rasterq4_kernel);
extern "C" int mw_rasterq_lds_bytes(int S, int W, int H, int n_tiles, int depth);
extern "C" int mw_rasterq_cap(int depth);
#define MW_RASTERQ_THREADS 512
extern "C" __global__ void mw_xcc_probe_kernel(uint32_t *out);
extern "C" __global__ void mw_mesh_entity_kernel(int N, int W, int H, const float *envhdr, const MwMeshDesc *meshes, const float4 *mesh_vpos, const uint2 *mesh_idx,
                                                 const float *mesh_stream, const float *mesh_attr, uint32_t *keys, float *plane_cache, int plane_cap,
                                                 int32_t *slow_count, uint32_t *slow_tris, const uint32_t *ent_list, int ent_list_cap, int32_t *ent_n, int32_t *ent_n_after,
                                                 uint32_t *slow_envs, int n_xcc, unsigned long long *prof);
extern "C" __global__ void mw_mesh_slow_kernel(int W, int H, const float *envhdr, const float *mesh_pos, const float *mesh_nrm, const float *mesh_rgb,
                                               const float *mesh_uv, const uint32_t *texels, int texel_bytes, uint32_t *keys, int32_t *counts, int N,
                                               int parity, const uint32_t *slow_tris, float4 *frags, uint32_t *heads, uint32_t stamp, uint32_t *status, const uint32_t *slow_envs, const int32_t *slow_env_n);
extern "C" __global__ void mw_reset_kernel(MwArgs a, const uint8_t *mask, int force_all, int mark_refill);
extern "C" __global__ void mw_refill_kernel(MwArgs a);
extern "C" __global__ void mw_refill_pcg_kernel(MwArgs a);
extern "C" __global__ void mw_collect_respawn_kernel(MwArgs a);
extern "C" __global__ void mw_collect_respawn_pcg_kernel(MwArgs a);
extern "C" __global__ void mw_take_spare_kernel(MwArgs a, const uint8_t *mask, int force_all);
extern "C" __global__ void mw_view_mesh_kernel(int W, int H, int S, int first_env, const float *envhdr, const float *mesh_pos, uint32_t *keys);
extern "C" __global__ void mw_view_raster_kernel(int env, int W, int H, int S, int max_vis, int tiles_x, const float *rec_raster,
                                                 const float *rec_shade, const float *rec_cull, const int32_t *nvis, const float *envhdr,
                                                 const MwTexDesc *texd, const uint32_t *texels, const float *mesh_pos,
                                                 const float *mesh_nrm, const float *mesh_rgb, const float *mesh_uv, const uint32_t *mesh_keys,
                                                 uint8_t *out, float *depth, int texel_bytes);
extern "C" __global__ void mw_visible_kernel(int env_base, int W, int H, int S, int max_vis, int E, const float *rec_raster, const float *rec_cull,
                                             const int32_t *nvis, uint8_t *vis);

#define MW_TIMING_STRIDE 8

// mw_get_info: what the envs' step() returns in `info` beside the observation (collecthealth.py:100, tmaze.py:89, ymaze.py:125)
extern "C" __global__ void mw_info_kernel(int N, int E, const int32_t *health, const double *epos, int slot, int32_t *out_health, double *out_pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (out_health) out_health[i] = health[i];
    if (out_pos)
        for (int c = 0; c < 3; ++c) out_pos[(size_t)i * 3 + c] = epos[((size_t)c * E + slot) * N + i];
}

namespace {
thread_local std::string g_create_error;
}

struct mw_engine {
    mw_config cfg{};
    MwArgs args{};
    MwArgs *d_gen_live = nullptr, *d_gen_spare = nullptr;   // device copies of the argument block for the generators
    bool spare_mode = false;
    bool side_refill_pending = false;   // Maze: spare worlds are regenerated by a kernel of their own on the side stream, across steps
    MwSpare spare_host{};
    int32_t *d_spare_dummy = nullptr;   // carry / step / picked written by the generator in spare mode go nowhere
    int n_sets = 1;
    std::string err;
    // device allocations (freed in destroy)
    std::vector<void *> allocs;
    // textures
    std::vector<MwTexDesc> tex_desc;
    std::vector<std::vector<uint32_t>> tex_data;   // per texture: every level as 32-byte footprint records (build_pyramid)
    uint32_t *d_texels = nullptr;
    MwTexDesc *d_texdesc = nullptr;
    MwMeshDesc *d_meshdesc = nullptr;
    std::vector<MwMeshDesc> mesh_desc;
    std::vector<std::vector<float>> mesh_pos, mesh_nrm, mesh_rgb, mesh_uv;   // per mesh id, [ntris][9] ([6] for uv)
    float *d_mesh_pos = nullptr, *d_mesh_nrm = nullptr, *d_mesh_rgb = nullptr, *d_mesh_uv = nullptr;
    float *d_mesh_stream = nullptr, *d_mesh_attr = nullptr;     // the entity kernel's triangle streams (rasterisation order): positions (meshes without a vertex table), vertex attributes
    float4 *d_mesh_vpos = nullptr;      // the meshes' distinct positions (MwMeshDesc::vfirst, nverts)
    uint2 *d_mesh_idx = nullptr;        // per triangle of the rasterisation order: three 16-bit indices into the mesh's table, the triangle's index
    std::vector<std::vector<float>> mesh_vtab;      // per mesh id: [nverts][4]
    std::vector<std::vector<uint32_t>> mesh_itab;   // per mesh id: [ntris][2]
    int max_mesh_verts = 0;
    bool have_meshes = false;
    uint32_t *d_view_keys = nullptr;    // sample keys of the generic-resolution path
    bool visible_attr_set = false;
    hipStream_t side_stream = nullptr;      // low priority: the Maze's spare-world refills beside the steps
    hipStream_t quad_stream = nullptr;      // low priority: the raster kernel's first part (every tile no mesh can touch) beside the mesh kernels
    hipEvent_t ev_mesh_fork = nullptr, ev_mesh_join = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t *d_mesh_keys = nullptr;    // [N][H][W][8] sample keys of the mesh scatter kernel (all-ones between frames)
    bool mesh_keys_dirty = true;
    int32_t *d_slow_count = nullptr;    // [2 parities][2][N] listed triangles, fragments
    int32_t *d_ent_counter = nullptr;   // [2][MW_CNT_WORDS] the work lists' lengths and cursors (mw_device.h: ent_list_n), this frame's and the next frame's
    uint32_t *d_slow_envs = nullptr;    // [2][N] the envs with triangles across a frustum plane (written by the entity kernel: the slow kernel's work list)
    uint32_t *d_tile_list = nullptr;    // [N * n_tiles] the mesh tiles' work list (written by the geometry kernel)
    static constexpr int mesh_tile_waves = 16384;       // wavefronts of the mesh tiles' launch, wavefront w taking the items w, w + 16384, ... of the list (4096: 139 us, 8192: 122, 16384: 112)
    uint32_t *d_ent_list = nullptr;     // [2][N * slots] the work list itself (written by the geometry kernel)
    int ent_list_cap = 0;
    static constexpr int ent_blocks = 512;      // its persistent workgroups: two of 512 lanes per CU (768 of them, or 256 of 1024 lanes: measured slower)
    uint32_t mesh_frame_seq = 1;
    uint32_t *d_slow_tris = nullptr;
    float4 *d_slow_frags = nullptr;
    uint32_t *d_slow_head = nullptr;
    float *d_plane_cache = nullptr;     // [N][plane_cap][16] + [N][plane_cap][4] attribute planes of the mesh triangles that win samples (mw_raster_mesh.hip)
    int plane_cap = 0, max_mesh_tris = 0;
    int obs_layout = MW_OBS_HWC_U8;
    size_t view_keys_bytes = 0;
    // scratch for the step outputs when the caller passes none
    float *d_reward_scratch = nullptr;
    uint8_t *d_flag_scratch = nullptr;
    int32_t *d_action_scratch = nullptr;
    uint8_t *d_mask = nullptr;
    double *d_step_override = nullptr;
    bool use_step_override = false;
    // timing
    bool timing = false;
    int timing_stride = MW_TIMING_STRIDE;
    uint64_t frame_count = 0;
    struct Ev { hipEvent_t a, b, c; };
    std::vector<Ev> ev_used, ev_free;
    int waves_per_env = 0;
    MwProgram *d_prog = nullptr;        // placement program (mw_set_gen_program)
    int texel_bytes = 4;
    int dbg_flags = 0;       // MW_DEBUG_FLAGS: perf experiments only (bit0: flat shading)
    int last_raster_path = -1;  // mw_raster_path
    // switches read once by mw_create (the launch path never touches the environment): the ones tests and A/B baselines use.
    // (The experiments that lost their A/B — the step fused into the geometry kernel, the quad kernel on big scenes, the stream
    // arrangements of the mesh kernels — are gone from the library: tools/experiments/ keeps the record and the patches.)
    bool use_k2q = true;        // MW_K2Q=0: the tile kernels of mw_raster.hip for small scenes too (the A/B baseline of the quad kernel)
    bool k2q_ok = false;        // the frame fits the quad kernel's LDS plan
    bool generic_raster = false;    // MW_GENERIC_RASTER=1: msaa = 4 frames through the generic-resolution kernel (tests run both)
    static constexpr int slow_waves = 8192;      // wavefronts of the slow kernel's launch (4096: 59 us, 8192: 55)
    unsigned long long *d_ent_prof = nullptr;   // MW_ENT_PROF=<file>: the mesh entity kernel's per-env times and counts of the last frame, [N][8], dumped by mw_destroy
    unsigned long long *d_k2q_prof = nullptr;   // MW_K2Q_PROF=<file>: s_memtime stamps of the quad kernel's phases, [N][8 waves][8], dumped by mw_destroy
};

namespace {

int fail(mw_engine *e, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(e, call)                                                                         \
    do {                                                                                         \
        hipError_t _st = (call);                                                                 \
        if (_st != hipSuccess)                                                                   \
            return fail(e, MW_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_st), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(mw_engine *e, T **out, size_t count, bool zero = true)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t st = hipMalloc(&p, bytes);
    if (st != hipSuccess) return fail(e, MW_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(st));
    if (zero) {
        st = hipMemset(p, 0, bytes);
        if (st != hipSuccess) return fail(e, MW_E_HIP, "hipMemset failed: %s", hipGetErrorString(st));
    }
    e->allocs.push_back(p);
    *out = static_cast<T *>(p);
    return MW_OK;
}

// K1 for the engine's random stream (the device code is compiled once per stream, mw_rng.h)
auto k1_of(const mw_engine *e) -> decltype(&mw_step_setup_kernel)
{
    return e->cfg.rng_mode == MW_RNG_PCG64 ? mw_step_setup_pcg_kernel : mw_step_setup_kernel;
}

int k1_threads(const mw_engine *) { return 64; }

// the geometry kernel: big scenes (one env per wavefront) or small, 8 samples per pixel (compiled in) or any
auto geom_kernel_of(const mw_engine *e, int L, int msaa) -> void (*)(MwArgs, int, int, int, int)
{
    const bool fixed8 = msaa == 8;
    if (L == 64) return fixed8 ? mw_geom_big_kernel : mw_geom_big_any_kernel;
    return fixed8 ? mw_geom_kernel : mw_geom_any_kernel;
}

// The tile / quad / mesh-scatter kernels keep edge values in 32 bits: |c_k| = |dcdx X - dcdy Y| <= 2 W H 2^16 has to stay below
// 2^31, i.e. W H < 16384 — 128 x 96 passes, 128 x 128 does not (a wall across the whole frame lost its triangle there);
// larger frames take the generic-resolution kernels (64-bit edge values).
bool tile_kernels_exact(int W, int H) { return W <= 128 && H <= 128 && W * H <= 128 * 96; }

// lanes per env of the geometry kernel: the power of two that holds an env's triangles (two per polygon and box face, the
// agent marker), 8 .. 64 — except that the smallest scenes get 16 lanes for their up to 32 triangles: an env's lanes go over
// its triangles in rounds, and four envs per wavefront fill the chip with half the wavefronts of this one-wave-per-SIMD kernel
// (measured, 4096 Hallway envs: 64 lanes 117 us, 32: 86, 16: 79, 8: 101)
int geom_lanes(const mw_engine *e)
{
    const int items = 2 * (e->cfg.max_polys + 6 * e->cfg.max_ents + 1);      // one triangle per lane
    int L = 8;
    while (L < items && L < 64) L <<= 1;
    if (L == 32) L = 16;
    // mid-sized scenes (PickupObjects: 6 polygons + 5 entity slots = 74 triangles; no visiting order, no sifting): two envs per
    // wavefront — 2 048 envs are ONE round of this one-wave-per-SIMD kernel instead of two (K1 + KG 103 -> 71 us)
    if (L == 64 && !e->args.rec_order && e->cfg.max_polys <= 64) L = 32;
#if defined(MW_PERF_HOOKS) || defined(MW_TUNE_HOOKS)
    if (const char *s = getenv("MW_GEOM_LANES")) { const int v = atoi(s); if ((v == 8 || v == 16 || v == 32 || v == 64) && v >= L) L = v; }
#endif
    return L;
}

// Lanes per env of the dense K1 (mw_setup_dense.hip), or 0 when the step has to go through the wave-per-env kernel: big
// scenes, CollectHealth, or too many slots to pack two envs into a wavefront.
int k1_dense_lanes(const mw_engine *e, int view_flags)
{
    if (e->args.rec_order || view_flags != 0 || e->cfg.task == MW_TASK_COLLECT) return 0;
    // at least two envs per wavefront: with one, every lane repeats the env's scalar work for nothing and the wave-per-env
    // kernel's lane-cooperative collision tests win (PickupObjects, 35 slots: 62 us dense against 47 us)
    const int lanes = e->cfg.max_polys + 6 * e->cfg.max_ents;
    return lanes <= 32 ? lanes : 0;
}

// numpy.random.SeedSequence(seed).generate_state(4, uint64) for a non-negative integer seed (the
// published SeedSequence algorithm: 4-word pool, hashmix / mix with the constants below), then PCG64's
// pcg_setseq_128_srandom_r — what gymnasium's np_random(seed) builds (miniworld.py:551).
void pcg64_seed(uint64_t seed, uint64_t out[4])
{
    const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
    const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_ent = ent[1] ? 2 : 1;
    uint32_t hc = INIT_A;
    auto hashmix = [&](uint32_t v) { v ^= hc; hc *= MULT_A; v *= hc; v ^= v >> 16; return v; };
    auto mix = [&](uint32_t x, uint32_t y) { uint32_t r = MIX_L * x - MIX_R * y; r ^= r >> 16; return r; };
    uint32_t pool[4];
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
    uint32_t hb = INIT_B, w[8];
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hb; hb *= MULT_B; v *= hb; v ^= v >> 16;
        w[i] = v;
    }
    uint64_t st[4];
    for (int i = 0; i < 4; ++i) st[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    // initstate = st[0]:st[1], initseq = st[2]:st[3];  inc = (initseq << 1) | 1
    const uint64_t inc_hi = (st[2] << 1) | (st[3] >> 63), inc_lo = (st[3] << 1) | 1ull;
    uint64_t hi = 0, lo = 0;
    mw::pcg64_step(hi, lo, inc_hi, inc_lo);
    const uint64_t sl = lo + st[1];
    hi += st[0] + (sl < lo ? 1ull : 0ull);
    lo = sl;
    mw::pcg64_step(hi, lo, inc_hi, inc_lo);
    out[0] = hi; out[1] = lo; out[2] = inc_hi; out[3] = inc_lo;
}

// (re)seed env i in a host copy of the uint64[4][N] rng array
void seed_env(const mw_engine *e, uint64_t *rng, int i, uint64_t seed)
{
    const size_t N = (size_t)e->cfg.num_envs;
    if (e->cfg.rng_mode == MW_RNG_PCG64) {
        uint64_t s[4];
        pcg64_seed(seed, s);
        for (int k = 0; k < 4; ++k) rng[(size_t)k * N + i] = s[k];
        rng[4 * N + i] = 0;
    } else {
        rng[i] = seed; rng[N + i] = 0; rng[2 * N + i] = 0; rng[3 * N + i] = 0; rng[4 * N + i] = 0;
    }
}

// Mip pyramid as glGenerateMipmap builds it on the reference's driver (llvmpipe: a GL_LINEAR blit of the previous level):
// destination texel i of dn reads source texels i0, i1 with an 8-bit weight — 24.8 fixed-point coordinate
// iround((i + 0.5) n / dn * 256) - 128, CLAMP_TO_EDGE; 2i, 2i + 1 with weight 128 on an even axis — and
// lerp a + ((w (b - a) + 128) >> 8), x first, then y.  tests/golden/gl_meta.npz holds the driver's own levels (checksums).
struct Taps { int i0, i1, w; };
Taps axis_taps(int n, int dn, int i)
{
    Taps t{0, 0, 0};
    if (n == 1) return t;
    const double sc = ((double)i + 0.5) * (double)n / (double)dn * 256.0;
    const long fixed = lrint(sc) - 128;         // round half to even
    const long ip = fixed >> 8;
    t.w = (int)(fixed & 255);
    t.i0 = ip < 0 ? 0 : (ip > n - 1 ? n - 1 : (int)ip);
    t.i1 = ip + 1 < 0 ? 0 : (ip + 1 > n - 1 ? n - 1 : (int)(ip + 1));
    return t;
}
inline int lerp8(int a, int b, int w) { return a + ((w * (b - a) + 128) >> 8); }

void build_pyramid(const uint8_t *rgb, int w, int h, std::vector<uint32_t> &out, MwTexDesc &desc)
{
    std::vector<uint8_t> cur(rgb, rgb + (size_t)w * h * 3), nxt;
    desc.w = (uint32_t)w; desc.h = (uint32_t)h; desc.nlevels = 0; desc.pad = 0;
    out.clear();
    for (;;) {
        // a level is stored as one 32-byte record per texel (i, j): its GL_LINEAR footprint (i, j), (i+1, j), (i, j+1),
        // (i+1, j+1), GL_REPEAT applied, laid out for the filter's first step.  The lerp along x of a channel's two texels
        // a, b under the 8-bit weight w, a + ((w (b - a) + 128) >> 8), is ((a * 256 + 128) + w * (b - a)) >> 8 in 16-bit
        // arithmetic (the sum stays in [128, 65408]); a record holds A = a * 256 + 128 and D = (b - a) mod 2^16, two
        // channels to a dword: row j as (A_r | A_b << 16, D_r | D_b << 16, A_g, D_g), then row j + 1 the same.  A bilinear
        // tap is two 16-byte loads, needs neither the neighbours' indices nor their wrap nor any unpacking, and its
        // x step is one packed multiply-add and one packed shift per pair of channels (8x the memory of the texels: the
        // coarse levels an 80x60 frame samples stay cache resident all the same).  Level::off counts records from the
        // start of the pool.
        desc.lvl[desc.nlevels++] = MwTexDesc::Level{(uint32_t)(out.size() / 8), (uint32_t)w, (uint32_t)w - 1u, (uint32_t)h - 1u, (float)w, (float)h, (uint32_t)h, 0u};
        auto chan = [&](int i, int j, int c) { return (uint32_t)cur[((size_t)(j % h) * w + (size_t)(i % w)) * 3 + c]; };
        auto A = [&](int i, int j, int c) { return chan(i, j, c) * 256u + 128u; };
        auto D = [&](int i, int j, int c) { return (chan(i + 1, j, c) - chan(i, j, c)) & 0xFFFFu; };
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i)
                for (int r = 0; r < 2; ++r) {
                    out.push_back(A(i, j + r, 0) | (A(i, j + r, 2) << 16)); out.push_back(D(i, j + r, 0) | (D(i, j + r, 2) << 16));
                    out.push_back(A(i, j + r, 1)); out.push_back(D(i, j + r, 1));
                }
        if ((w == 1 && h == 1) || desc.nlevels == MW_MAX_LEVELS) break;
        const int nw = std::max(1, w / 2), nh = std::max(1, h / 2);
        nxt.assign((size_t)nw * nh * 3, 0);
        for (int j = 0; j < nh; ++j) {
            const Taps ty = axis_taps(h, nh, j);
            for (int i = 0; i < nw; ++i) {
                const Taps tx = axis_taps(w, nw, i);
                for (int c = 0; c < 3; ++c) {
                    const int t0 = lerp8(cur[((size_t)ty.i0 * w + tx.i0) * 3 + c], cur[((size_t)ty.i0 * w + tx.i1) * 3 + c], tx.w);
                    const int t1 = lerp8(cur[((size_t)ty.i1 * w + tx.i0) * 3 + c], cur[((size_t)ty.i1 * w + tx.i1) * 3 + c], tx.w);
                    nxt[((size_t)j * nw + i) * 3 + c] = (uint8_t)lerp8(t0, t1, ty.w);
                }
            }
        }
        cur.swap(nxt);
        w = nw; h = nh;
    }
}

int sync_gen_args(mw_engine *e);

// One device block holds the descriptor table followed by the texels of every level: the raster kernels reach
// both through a single buffer resource (4 SGPRs instead of 8), texel offsets count dwords from the block's start.
int upload_textures(mw_engine *e)
{
    const size_t table = (size_t)MW_MAX_TEX * sizeof(MwTexDesc) / 4;       // dwords
    size_t total = table;
    std::vector<MwTexDesc> descs = e->tex_desc;
    static_assert((MW_MAX_TEX * sizeof(MwTexDesc)) % 32 == 0, "footprint records are 32-byte aligned behind the table");
    for (size_t i = 0; i < descs.size(); ++i) {
        for (uint32_t l = 0; l < descs[i].nlevels; ++l) descs[i].lvl[l].off += (uint32_t)(total / 8);      // in 32-byte records
        total += e->tex_data[i].size();
    }
    if (total * 4 > 0xFFFFFFF0ull) return fail(e, MW_E_CAPACITY, "texture pool of %zu bytes exceeds one buffer resource", total * 4);
    if (e->d_texels) { (void)hipFree(e->d_texels); e->d_texels = nullptr; e->d_texdesc = nullptr; }
    HIP_TRY(e, hipMalloc((void **)&e->d_texels, total * 4));
    e->d_texdesc = reinterpret_cast<MwTexDesc *>(e->d_texels);
    size_t off = table;
    for (size_t i = 0; i < descs.size(); ++i) {
        if (!e->tex_data[i].empty())
            HIP_TRY(e, hipMemcpy(e->d_texels + off, e->tex_data[i].data(), e->tex_data[i].size() * 4, hipMemcpyHostToDevice));
        off += e->tex_data[i].size();
    }
    HIP_TRY(e, hipMemcpy(e->d_texdesc, descs.data(), descs.size() * sizeof(MwTexDesc), hipMemcpyHostToDevice));
    e->args.texels = e->d_texels;
    e->args.tex = e->d_texdesc;
    e->texel_bytes = (int)(total * 4);
    if (e->d_gen_live && sync_gen_args(e) != MW_OK) return MW_E_HIP;
    return MW_OK;
}

int pick_waves_per_env(const mw_engine *e)
{
    const int n_tiles = e->args.n_tiles;
    // enough wavefronts to fill 256 CUs x 4 SIMDs x 7 resident waves several times over (measured:
    // 15-25 waves per env beat 5 by ~7 % at 4096 envs), in divisors of n_tiles
    int best = n_tiles;
#if defined(MW_PERF_HOOKS) || defined(MW_TUNE_HOOKS)      // (MW_TUNE_HOOKS: the launch-shape overrides alone, without the perf build's counters)
    if (const char *s = getenv("MW_WAVES_PER_ENV")) { const int v = atoi(s); if (v > 0 && n_tiles % v == 0) return v; }
#endif
    for (int w = 1; w <= n_tiles; ++w) {
        if (n_tiles % w) continue;
        if ((long long)e->cfg.num_envs * w >= 49152) { best = w; break; }
    }
    return best;
}

// copy host [count][inner] <-> device SoA [inner][N] (component-major), element type T
template <typename T>
int xfer(mw_engine *e, T *dev, T *host, int first, int count, int inner, bool to_device)
{
    if (!host) return MW_OK;
    const int N = e->cfg.num_envs;
    std::vector<T> tmp((size_t)count);
    for (int k = 0; k < inner; ++k) {
        T *d = dev + (size_t)k * N + first;
        if (to_device) {
            for (int i = 0; i < count; ++i) tmp[i] = host[(size_t)i * inner + k];
            HIP_TRY(e, hipMemcpy(d, tmp.data(), sizeof(T) * count, hipMemcpyHostToDevice));
        } else {
            HIP_TRY(e, hipMemcpy(tmp.data(), d, sizeof(T) * count, hipMemcpyDeviceToHost));
            for (int i = 0; i < count; ++i) host[(size_t)i * inner + k] = tmp[i];
        }
    }
    return MW_OK;
}

// host [count][E][inner] <-> device [inner][E][N]
template <typename T>
int xfer_ent(mw_engine *e, T *dev, T *host, int first, int count, int inner, bool to_device)
{
    if (!host) return MW_OK;
    const int N = e->cfg.num_envs, E = e->cfg.max_ents;
    std::vector<T> tmp((size_t)count);
    for (int k = 0; k < inner; ++k)
        for (int s = 0; s < E; ++s) {
            T *d = dev + ((size_t)k * E + s) * N + first;
            if (to_device) {
                for (int i = 0; i < count; ++i) tmp[i] = host[((size_t)i * E + s) * inner + k];
                HIP_TRY(e, hipMemcpy(d, tmp.data(), sizeof(T) * count, hipMemcpyHostToDevice));
            } else {
                HIP_TRY(e, hipMemcpy(tmp.data(), d, sizeof(T) * count, hipMemcpyDeviceToHost));
                for (int i = 0; i < count; ++i) host[((size_t)i * E + s) * inner + k] = tmp[i];
            }
        }
    return MW_OK;
}

int state_xfer(mw_engine *e, int first, int count, const mw_state_view *h, bool to_device)
{
    if (!e || !h) return fail(e, MW_E_INVALID, "null argument");
    if (first < 0 || count < 0 || first + count > e->cfg.num_envs) return fail(e, MW_E_INVALID, "env range out of bounds");
    MwArgs &a = e->args;
    int rc;
    // agent_pos is [count][3] on the host, three separate arrays on the device
    if (h->agent_pos) {
        std::vector<double> tmp((size_t)count);
        double *dev[3] = {a.ax, a.ay, a.az};
        for (int k = 0; k < 3; ++k) {
            if (to_device) {
                for (int i = 0; i < count; ++i) tmp[i] = h->agent_pos[(size_t)i * 3 + k];
                HIP_TRY(e, hipMemcpy(dev[k] + first, tmp.data(), 8 * (size_t)count, hipMemcpyHostToDevice));
            } else {
                HIP_TRY(e, hipMemcpy(tmp.data(), dev[k] + first, 8 * (size_t)count, hipMemcpyDeviceToHost));
                for (int i = 0; i < count; ++i) h->agent_pos[(size_t)i * 3 + k] = tmp[i];
            }
        }
    }
    if ((rc = xfer(e, a.adir, h->agent_dir, first, count, 1, to_device))) return rc;
    if ((rc = xfer(e, a.cam, h->cam, first, count, 4, to_device))) return rc;
    if ((rc = xfer(e, a.light, h->light, first, count, 12, to_device))) return rc;
    if ((rc = xfer(e, a.carry, h->carrying, first, count, 1, to_device))) return rc;
    if ((rc = xfer(e, a.step, h->step_count, first, count, 1, to_device))) return rc;
    if ((rc = xfer(e, a.picked, h->num_picked_up, first, count, 1, to_device))) return rc;
    if ((rc = xfer_ent(e, a.ekind, h->ent_kind, first, count, 1, to_device))) return rc;
    if ((rc = xfer_ent(e, a.emesh, h->ent_mesh, first, count, 1, to_device))) return rc;
    if ((rc = xfer_ent(e, a.estatic, h->ent_static, first, count, 1, to_device))) return rc;
    if ((rc = xfer_ent(e, a.epos, h->ent_pos, first, count, 3, to_device))) return rc;
    if ((rc = xfer_ent(e, a.edir, h->ent_dir, first, count, 1, to_device))) return rc;
    if ((rc = xfer_ent(e, a.egeom, h->ent_geom, first, count, 9, to_device))) return rc;
    if ((rc = xfer(e, a.extent, h->extent, first, count, 4, to_device))) return rc;
    return MW_OK;
}

// every entry point runs on the engine's device, whatever the calling thread's current device is (two engines
// on different GPUs in one process; torch's current device != cfg.device_id)
#define ON_DEVICE(e) do { hipError_t sd_ = hipSetDevice((e)->cfg.device_id); \
        if (sd_ != hipSuccess) return fail((e), MW_E_HIP, "hipSetDevice(%d): %s", (e)->cfg.device_id, hipGetErrorString(sd_)); } while (0)

// ... and, for every entry point but the step / render ones, after the spare-world refills still running on the side stream
#define ON_DEVICE_SYNC(e) do { ON_DEVICE(e); if ((e)->side_refill_pending) { (void)hipStreamSynchronize((e)->side_stream); \
        (e)->side_refill_pending = false; } } while (0)

mw_engine::Ev get_events(mw_engine *e)
{
    if (!e->ev_free.empty()) {
        mw_engine::Ev ev = e->ev_free.back();
        e->ev_free.pop_back();
        return ev;
    }
    mw_engine::Ev ev{};
    (void)hipEventCreate(&ev.a);
    (void)hipEventCreate(&ev.b);
    (void)hipEventCreate(&ev.c);
    return ev;
}

// Device copies of the argument block for the generators (live state; spare state with the world pointers
// redirected): generate_world indexes the block dynamically, which a by-value kernarg would turn into a scratch copy.
int sync_gen_args(mw_engine *e)
{
    if (e->cfg.generator == MW_GEN_NONE) return MW_OK;
    if (!e->d_gen_live) {
        HIP_TRY(e, hipMalloc((void **)&e->d_gen_live, sizeof(MwArgs)));
        e->allocs.push_back(e->d_gen_live);
        e->args.gen_live = e->d_gen_live;
        if (e->spare_mode) {
            HIP_TRY(e, hipMalloc((void **)&e->d_gen_spare, sizeof(MwArgs)));
            e->allocs.push_back(e->d_gen_spare);
            e->args.gen_spare = e->d_gen_spare;
        }
    }
    MwArgs live = e->args;
    HIP_TRY(e, hipMemcpy(e->d_gen_live, &live, sizeof live, hipMemcpyHostToDevice));
    if (e->spare_mode) {
        // everything of the world goes to the spare arrays; the random stream (rng) and the status word stay the live ones
        MwArgs sa = e->args;
        const MwSpare &sp = e->spare_host;
        sa.ax = sp.ax; sa.ay = sp.ay; sa.az = sp.az; sa.adir = sp.adir; sa.cam = sp.cam; sa.light = sp.light; sa.extent = sp.extent;
        sa.ekind = sp.ekind; sa.emesh = sp.emesh; sa.estatic = sp.estatic; sa.epos = sp.epos; sa.edir = sp.edir; sa.egeom = sp.egeom;
        sa.carry = e->d_spare_dummy; sa.step = e->d_spare_dummy + e->cfg.num_envs; sa.picked = e->d_spare_dummy + 2 * (size_t)e->cfg.num_envs;
        if (!e->cfg.shared_geometry) { sa.polys = sp.polys; sa.npolys = sp.npolys; sa.segs = sp.segs; sa.nsegs = sp.nsegs; sa.occ_valid = nullptr; sa.occ_cache = nullptr; }
        sa.spare = nullptr;
        HIP_TRY(e, hipMemcpy(e->d_gen_spare, &sa, sizeof sa, hipMemcpyHostToDevice));
    }
    return MW_OK;
}

// second, low-priority stream for work that runs beside the raster kernel (spare refill, K2 beside the mesh kernel)
int ensure_side_stream(mw_engine *e)
{
    if (e->side_stream) return MW_OK;
    int prio_least = 0, prio_greatest = 0;      // the filler work must not keep the main kernels' workgroups out
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIP_TRY(e, hipStreamCreateWithPriority(&e->side_stream, hipStreamNonBlocking, prio_least));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    return MW_OK;
}

// the stream of the raster kernel's first part in a frame with meshes: LOW priority — the mesh kernels on the caller's stream are the
// critical path, the quad kernel fills the CUs around them
int ensure_mesh_stream(mw_engine *e)
{
    if (e->quad_stream) return MW_OK;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIP_TRY(e, hipStreamCreateWithPriority(&e->quad_stream, hipStreamNonBlocking, prio_least));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_mesh_fork, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_mesh_join, hipEventDisableTiming));
    return MW_OK;
}

// Everything a frame with mesh entities needs beyond the triangle records — the plane cache (one record per mesh triangle that
// can be in view: the geometry kernel admits 0xC000 per env), the sample keys of the tiles a mesh can touch, the slow-path
// lists, the mesh stream; for the generic-resolution path the view keys.  Called by mw_upload_mesh (a synchronous entry point):
// a frame never allocates, never synchronises.  Failure-atomic: either every buffer of a group is there or none.
int ensure_mesh_buffers(mw_engine *e)
{
    const MwArgs &a = e->args;
    const size_t N = (size_t)e->cfg.num_envs;
    if (ensure_mesh_stream(e) != MW_OK) return MW_E_HIP;
    if (e->cfg.msaa != 8 || !tile_kernels_exact(a.W, a.H)) {       // the generic-resolution path
        const size_t need = N * a.W * a.H * e->cfg.msaa * 4;
        if (need > e->view_keys_bytes) {
            uint32_t *nk = nullptr;
            if (hipMalloc((void **)&nk, need) != hipSuccess) return fail(e, MW_E_NOMEM, "hipMalloc(%zu) for the view keys failed", need);
            (void)hipDeviceSynchronize();
            if (e->d_view_keys) (void)hipFree(e->d_view_keys);
            e->d_view_keys = nk; e->view_keys_bytes = need;
        }
        return MW_OK;
    }
    if (a.W > 255 * MW_TILE_W || a.H > 255 * MW_TILE_H) return fail(e, MW_E_CAPACITY, "frame too large for the mesh tile rectangles");
    // the mesh tiles' work list (mw_geom.hip): a tile index in the 8 bits above the env, and one bit of a lane's 32-bit mask per tile
    // sub + k L.  tile_kernels_exact caps these frames at 192 tiles and the geometry kernel has at least 8 lanes per env, so this holds
    // today; a larger frame limit or fewer lanes must not leave mesh tiles undrawn (and their sample keys uncleared) in silence
    if (a.n_tiles > 255 || a.n_tiles > 32 * geom_lanes(e))
        return fail(e, MW_E_CAPACITY, "%d tiles per frame: the mesh tiles' work list holds 255 (8-bit tile index) and 32 per lane of the geometry kernel (%d lanes)", a.n_tiles, geom_lanes(e));
    const long long want = std::min<long long>(0xC000, (long long)e->cfg.max_ents * e->max_mesh_tris);
    if ((int)want > e->plane_cap) {
        float *np = nullptr;
        if (hipMalloc((void **)&np, N * (size_t)want * (MW_PLANE_REC + MW_PLANE_XTRA) * 4) != hipSuccess) return fail(e, MW_E_NOMEM, "hipMalloc for the plane cache (%lld records per env) failed", want);
        (void)hipDeviceSynchronize();
        if (e->d_plane_cache) (void)hipFree(e->d_plane_cache);
        e->d_plane_cache = np; e->plane_cap = (int)want;
    }
    if (!e->d_ent_list) {
        // (all three work lists or none: a failed allocation leaves no pointer behind)
        const int cap = (int)N * std::min(MW_MAX_MESH_ENTS, std::max(e->cfg.max_ents, 1));
        void *ents = nullptr, *slow = nullptr, *tiles = nullptr;
        const bool ok = hipMalloc(&ents, (size_t)cap * 16 * 4) == hipSuccess && hipMalloc(&slow, N * 2 * 4) == hipSuccess &&
                        hipMalloc(&tiles, N * (size_t)a.n_tiles * 8 * 4) == hipSuccess;
        if (!ok) {
            for (void *p : {ents, slow, tiles}) if (p) (void)hipFree(p);
            return fail(e, MW_E_NOMEM, "hipMalloc for the mesh path's work lists failed");
        }
        e->ent_list_cap = cap;
        e->d_ent_list = (uint32_t *)ents; e->d_slow_envs = (uint32_t *)slow; e->d_tile_list = (uint32_t *)tiles;
    }
    if (!e->d_mesh_keys) {
        const size_t key_bytes = N * a.W * a.H * 8 * 4, head_bytes = N * a.W * a.H * 4;
        void *keys = nullptr, *cnt = nullptr, *tris = nullptr, *frags = nullptr, *head = nullptr;
        // triangles that cross a frustum plane and their fragments (mw_mesh_slow_kernel): counts, 1024 / 2048 entries per env
        const bool ok = hipMalloc(&keys, key_bytes) == hipSuccess && hipMalloc(&cnt, N * 4 * 4 + 2 * MW_CNT_WORDS * 4) == hipSuccess && hipMalloc(&tris, N * MW_SLOW_TRIS * 4) == hipSuccess &&
                        hipMalloc(&frags, N * MW_SLOW_STRIDE * 16) == hipSuccess && hipMalloc(&head, head_bytes) == hipSuccess &&
                        hipMemset(cnt, 0, N * 4 * 4 + 2 * MW_CNT_WORDS * 4) == hipSuccess && hipMemset(head, 0, head_bytes) == hipSuccess && hipMemset(keys, 0xFF, key_bytes) == hipSuccess;
        if (!ok) {
            for (void *p : {keys, cnt, tris, frags, head}) if (p) (void)hipFree(p);
            return fail(e, MW_E_NOMEM, "hipMalloc for the mesh path's buffers failed");
        }
        e->d_mesh_keys = (uint32_t *)keys; e->d_slow_count = (int32_t *)cnt; e->d_slow_tris = (uint32_t *)tris;
        e->d_slow_frags = (float4 *)frags; e->d_slow_head = (uint32_t *)head;
        e->d_ent_counter = (int32_t *)cnt + N * 4;       // (behind the slow path's counts)
        // (the memsets above ran on the null stream, which a caller's non-blocking stream is not ordered against: finish them here)
        (void)hipDeviceSynchronize();
        e->mesh_keys_dirty = false;
    }
    return MW_OK;
}

int launch_frame(mw_engine *e, bool do_step, int view_flags, const int32_t *d_actions, uint8_t *d_obs, float *d_depth,
                 float *d_reward, uint8_t *d_term, uint8_t *d_trunc, hipStream_t st)
{
    if (!d_obs) return fail(e, MW_E_INVALID, "d_obs is null");
    // (checked before anything is launched or any timing event is taken)
    if ((e->cfg.msaa != 8 || !tile_kernels_exact(e->cfg.obs_width, e->cfg.obs_height)) && e->obs_layout != MW_OBS_HWC_U8)
        return fail(e, MW_E_INVALID, "wrapper layouts need msaa = 8 and observations up to 128 x 96");
    MwArgs a = e->args;
    a.step_override = e->use_step_override ? e->d_step_override : nullptr;
    const int N = e->cfg.num_envs;
    // a frame with mesh entities through the tile / quad kernels: the geometry kernel lists the entities in view for the mesh
    // entity kernel (lists, slow-path lists and fragment stamps alternate between two sets from frame to frame)
    const bool mesh_obs = e->have_meshes && e->cfg.msaa == 8 && tile_kernels_exact(a.W, a.H) && e->d_ent_list && e->d_mesh_keys;
    const uint32_t mesh_seq = mesh_obs ? e->mesh_frame_seq++ : 0u;
    if (mesh_obs) {
        a.ent_list = e->d_ent_list; a.ent_list_n = e->d_ent_counter + (mesh_seq & 1u) * MW_CNT_WORDS; a.ent_list_cap = e->ent_list_cap;
        a.tile_list = e->d_tile_list; a.tile_list_cap = N * a.n_tiles;
    }
    mw_engine::Ev ev{};
    // kernel durations are sampled: three event records on every launch cost ~4 % of the step rate,
    // on one launch in MW_TIMING_STRIDE they cost nothing measurable
    const bool timed = e->timing && (e->frame_count++ % (uint64_t)e->timing_stride) == 0;
    if (timed) {
        ev = get_events(e);
        (void)hipEventRecord(ev.a, st);
    }
    // spare mode: blocks appended to the grid regenerate the spare worlds consumed in earlier steps, beside the step itself
    // ... except for the Maze: regenerating one takes ~300 us on a single wave, four times a whole step of the batch, and
    // any launch that carries such a block lasts that long.  Its refills go to a kernel of their own on the low-priority
    // side stream (below), running beside this and the next steps; nothing waits for it but the entry points that touch
    // the worlds from the host (ON_DEVICE_SYNC) — an env that needs its spare earlier follows the refill_mask protocol.
    const bool async_refill = e->spare_mode && do_step && e->cfg.generator == MW_GEN_MAZE;
    const int refill_blocks = (e->spare_mode && do_step && !async_refill) ? (N + 63) / 64 : 0;
    const int gl = geom_lanes(e);
    if (!do_step) {
        // render only: nothing to step
    } else if (const int lanes = k1_dense_lanes(e, 0)) {
        const int epw = 64 / lanes;
        const bool pcg = e->cfg.rng_mode == MW_RNG_PCG64;
        auto k1d = pcg ? mw_step_setup_dense_pcg_kernel : mw_step_setup_dense_kernel;
        // spare mode: blocks appended to the grid regenerate the spare worlds consumed in earlier steps (64 envs each)
        const int refill = (e->spare_mode && do_step) ? (N + 63) / 64 : 0;
        hipLaunchKernelGGL(k1d, dim3((N + epw - 1) / epw + refill), dim3(64), 0, st, a, do_step ? 1 : 0, lanes, d_actions,
                           d_reward ? d_reward : e->d_reward_scratch, d_term ? d_term : e->d_flag_scratch,
                           d_trunc ? d_trunc : e->d_flag_scratch + N);
    } else {
        hipLaunchKernelGGL(k1_of(e), dim3(N + refill_blocks), dim3(k1_threads(e)), 0, st, a, do_step ? 1 : 0, view_flags, d_actions,
                           d_reward ? d_reward : e->d_reward_scratch, d_term ? d_term : e->d_flag_scratch,
                           d_trunc ? d_trunc : e->d_flag_scratch + N);
    }
    // the frame's vertex half: camera, lighting, transform, clipping, triangle setup (mw_geom.hip)
    {
        const int L = gl, epw = 64 / L;
        hipLaunchKernelGGL(geom_kernel_of(e, L, e->cfg.msaa), dim3((N + epw - 1) / epw), dim3(64), 0, st, a, view_flags, e->cfg.msaa, L, N);
    }
    if (do_step && e->cfg.task == MW_TASK_COLLECT)
        hipLaunchKernelGGL(e->cfg.rng_mode == MW_RNG_PCG64 ? mw_collect_respawn_pcg_kernel : mw_collect_respawn_kernel, dim3((N + 63) / 64), dim3(64), 0, st, a);
    if (timed) (void)hipEventRecord(ev.b, st);
    if (async_refill) {
        if (ensure_side_stream(e) != MW_OK) return MW_E_HIP;
        HIP_TRY(e, hipEventRecord(e->ev_fork, st));
        HIP_TRY(e, hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
        hipLaunchKernelGGL(e->cfg.rng_mode == MW_RNG_PCG64 ? mw_refill_pcg_kernel : mw_refill_kernel, dim3(N), dim3(64), 0, e->side_stream, e->args);
        e->side_refill_pending = true;
    }
    // the quad kernel (mw_rasterq.hip): small scenes without a visiting order, frames that fit its LDS plan — 8 samples (the
    // hot path) and 4 (llvmpipe's GL_MAX_SAMPLES: the reference's own frames run through the same code); with mesh entities
    // it draws the tiles no mesh can touch (8 samples only)
    const bool big_scene = a.rec_order != nullptr;
    // (big scenes — a visiting order exists — keep the tile kernels: their near-to-far order with its early exit is the better fit
    // for deep scenes; the quad kernel on the Maze was measured and lost, tools/experiments/README.md)
    const bool k2q = e->use_k2q && e->k2q_ok && !big_scene && !(e->cfg.msaa == 4 && (e->have_meshes || e->generic_raster));
    auto launch_k2q = [&](int part_flags, hipStream_t kq) {
        const int S = e->cfg.msaa;
        const int lds = mw_rasterq_lds_bytes(S, a.W, a.H, a.n_tiles, d_depth ? 1 : 0);
        const int flags = (e->dbg_flags & 0xFC8F) | (e->obs_layout << 8) | part_flags;
        hipLaunchKernelGGL(S == 8 ? mw_rasterq_kernel : mw_rasterq4_kernel, dim3(N), dim3(MW_RASTERQ_THREADS), (size_t)lds, kq, a.N, a.W, a.H, a.max_vis,
                           a.tiles_x, a.n_tiles, (const float *)a.rec_raster, (const float *)a.rec_shade, (const float *)a.rec_cull,
                           (const int32_t *)a.nvis, (const float *)a.envhdr, a.texels, d_obs, d_depth, flags, e->texel_bytes, e->d_k2q_prof);
    };
    if (k2q && e->cfg.msaa == 4) {
        launch_k2q(0, st);
        e->last_raster_path = MW_PATH_QUAD;
    } else if (e->cfg.msaa != 8 || !tile_kernels_exact(a.W, a.H)) {
        // FrameBuffer's fallback sample counts (opengl.py:229-231: a driver that clamps GL_MAX_SAMPLES gets 4 or 1
        // samples) and observations beyond 128 x 128 (the tile kernels' 24-bit edge arithmetic): not the hot path — the
        // generic-resolution kernels, 64-bit edge values, exact packed-key resolution, the whole batch in one grid
        // (blockIdx.y = env)
        const int S = e->cfg.msaa;
        uint32_t *keys = nullptr;
        if (e->have_meshes) {
            const size_t need = (size_t)N * a.W * a.H * S * 4;
            if (need > e->view_keys_bytes) return fail(e, MW_E_INVALID, "view keys missing (mw_upload_mesh allocates them)");
            keys = e->d_view_keys;
            HIP_TRY(e, hipMemsetAsync(keys, 0xFF, need, st));
            hipLaunchKernelGGL(mw_view_mesh_kernel, dim3(32, N), dim3(256), 0, st, a.W, a.H, S, 0, (const float *)a.envhdr, a.mesh_pos, keys);
        }
        e->last_raster_path = MW_PATH_GENERIC;
        hipLaunchKernelGGL(mw_view_raster_kernel, dim3(a.n_tiles, N), dim3(64), 0, st, 0, a.W, a.H, S, a.max_vis, a.tiles_x,
                           (const float *)a.rec_raster, (const float *)a.rec_shade, (const float *)a.rec_cull, (const int32_t *)a.nvis, (const float *)a.envhdr,
                           a.tex, a.texels, a.mesh_pos, a.mesh_nrm, a.mesh_rgb, a.mesh_uv, (const uint32_t *)keys, d_obs, d_depth, e->texel_bytes);
    } else {
        const bool mesh = e->have_meshes;
        uint32_t mesh_stamp = 0u;
        if (mesh) {
            // (plane cache, sample keys — all-ones between frames, K2 clears what it reads —, slow-path lists, mesh stream:
            // ensure_mesh_buffers, at upload time)
            if (!e->d_mesh_keys || !e->d_plane_cache || !e->quad_stream) return fail(e, MW_E_INVALID, "mesh buffers missing (mw_upload_mesh allocates them)");
            const size_t key_bytes = (size_t)N * a.W * a.H * 8 * 4;
            if (e->mesh_keys_dirty) HIP_TRY(e, hipMemsetAsync(e->d_mesh_keys, 0xFF, key_bytes, st));
            e->mesh_keys_dirty = true;      // until the raster kernel that clears them again has been enqueued
            // frame stamp of the slow-fragment chains and parity of the lists.  The stamp has 16 bits: a head that no frame has
            // overwritten since frame F would read as valid again at frame F + 65536 (24 s of PickupObjects), so the heads are
            // wiped on the frame whose stamp is 0 — behind the previous frame's readers, before this frame's slow-path kernel, in
            // stream order (tests/test_gpu_env_api.py::test_slow_fragment_heads_survive_the_stamp_wrap)
            const uint32_t seq = mesh_seq;
            mesh_stamp = seq & 0xFFFFu;
            if (mesh_stamp == 0u) HIP_TRY(e, hipMemsetAsync(e->d_slow_head, 0, (size_t)N * a.W * a.H * 4, st));
            const int parity = (int)(seq & 1u);
            // The mesh kernels — the frame's critical path — stay on the caller's stream, right behind the geometry kernel; the quad
            // kernel, which draws every tile no mesh can touch, goes to the low-priority quad stream beside them.  (The other way
            // round — mesh kernels on a side stream — the quad kernel started a few microseconds EARLIER, its 2 048 workgroups
            // took the CUs, and the entity kernel's workgroups waited a quad-kernel workgroup's lifetime for room: 212 instead of
            // 133 us, PickupObjects 4.55 -> 5.3 M env-steps/s.)
            HIP_TRY(e, hipEventRecord(e->ev_mesh_fork, st));
            HIP_TRY(e, hipStreamWaitEvent(e->quad_stream, e->ev_mesh_fork, 0));
            // persistent workgroups drawing entities from the geometry kernel's list (two sets of counters swapping places: the
            // kernel zeroes the next frame's)
            hipLaunchKernelGGL(mw_mesh_entity_kernel, dim3(std::max(a.n_xcc, std::min(e->ent_list_cap, e->ent_blocks))), dim3(MW_ENT_THREADS), (size_t)e->max_mesh_verts * 16, st, N, a.W, a.H,
                               (const float *)a.envhdr, a.mesh, (const float4 *)e->d_mesh_vpos, (const uint2 *)e->d_mesh_idx, (const float *)e->d_mesh_stream,
                               (const float *)e->d_mesh_attr, e->d_mesh_keys, e->d_plane_cache, e->plane_cap, e->d_slow_count + (size_t)parity * 2 * N, e->d_slow_tris,
                               (const uint32_t *)e->d_ent_list, e->ent_list_cap, e->d_ent_counter + parity * MW_CNT_WORDS, e->d_ent_counter + (parity ^ 1) * MW_CNT_WORDS, e->d_slow_envs + (size_t)parity * N, e->args.n_xcc, e->d_ent_prof);
            hipLaunchKernelGGL(mw_mesh_slow_kernel, dim3(std::min(N * 16, e->slow_waves)), dim3(64), 0, st, a.W, a.H, (const float *)a.envhdr, a.mesh_pos, a.mesh_nrm, a.mesh_rgb,
                               a.mesh_uv, a.texels, e->texel_bytes, e->d_mesh_keys, e->d_slow_count, N, parity, (const uint32_t *)e->d_slow_tris,
                               e->d_slow_frags, e->d_slow_head, mesh_stamp, a.status,
                               (const uint32_t *)(e->d_slow_envs + (size_t)parity * N), (const int32_t *)(e->d_ent_counter + parity * MW_CNT_WORDS + MW_CNT_SLOW_ENVS));
        }
        const int wpe = e->waves_per_env;
        const int tpw = (a.n_tiles + wpe - 1) / wpe;
        const int groups = (N + 7) / 8;
        // big scenes (a visiting order exists): records read in place, near to far; otherwise the env's records are staged
        // in LDS when there are at most MW_LDS_RECS of them (a wave whose env holds more reads them in place).
        const bool big = a.rec_order != nullptr;
        const int lds_recs = a.max_vis < MW_LDS_RECS ? a.max_vis : MW_LDS_RECS;
        const size_t lds = big ? 192 : (size_t)lds_recs * (MW_LDS_SHADE_Q + MW_LDS_CULL_Q) * 16 + 192;
        // small scenes: the production kernels carry neither debug flags nor a run-time depth switch (mw_raster.hip);
        // anything else goes to the general kernel
        const bool general = e->obs_layout != MW_OBS_HWC_U8 || e->dbg_flags != 0;
        auto k2 = big ? (d_depth ? mw_raster_big_depth_kernel : mw_raster_big_kernel) : (d_depth ? mw_raster_depth_kernel : mw_raster_kernel);
        if (general) k2 = big ? mw_raster_big_wrap_kernel : mw_raster_wrap_kernel;
        if (mesh) {
            k2 = big ? mw_raster_big_mesh_wrap_kernel : (d_depth ? mw_raster_mesh_depth_kernel : mw_raster_mesh_kernel);
            if (general && !big) k2 = mw_raster_mesh_wrap_kernel;
        }
        const int flags = e->dbg_flags | (e->obs_layout << 8) | (int)(mesh_stamp << 16);
        // K2's first part of a frame with meshes never enters a mesh tile: the plain tile code with the skip (the small-scene observation path only)
        auto k2_first = (mesh && !big && !general) ? (d_depth ? mw_raster_nomesh_depth_kernel : mw_raster_nomesh_kernel) : k2;
        auto launch_k2 = [&](int part_flags, hipStream_t ks) {
            // the second part (the tiles a mesh can touch: few, slow, clustered): persistent wavefronts over the geometry kernel's
            // tile list (flags 3 << 4)
            const bool listed = (part_flags >> 4) == 2 && a.tile_list != nullptr;
            if (listed) part_flags = (part_flags & ~0x30) | (3 << 4);
            const int wpe2 = (part_flags >> 4) == 2 ? a.n_tiles : wpe;
            const int tpw2 = (part_flags >> 4) == 2 ? 1 : tpw;
            const int grid = listed ? std::min(e->mesh_tile_waves, N * (int)a.n_tiles) : groups * 8 * wpe2;
            hipLaunchKernelGGL((part_flags >> 4) == 1 ? k2_first : k2, dim3(grid), dim3(64), lds, ks, a.N, a.W, a.H, a.max_vis, a.tiles_x,
                               a.n_tiles, wpe2, tpw2, (const float *)a.rec_raster, (const float *)a.rec_shade, (const float *)a.rec_cull,
                               (const int32_t *)a.nvis,
                               (const float *)a.envhdr, a.tex, a.texels, d_obs, d_depth, flags | part_flags, e->texel_bytes,
                               (const uint16_t *)a.rec_order, a.mesh_pos, a.mesh_nrm, a.mesh_rgb, a.mesh_uv, e->d_mesh_keys,
                               (const float *)e->d_plane_cache, e->plane_cap, (const float4 *)e->d_slow_frags, (const uint32_t *)e->d_slow_head,
                               (const uint32_t *)a.tile_list, a.ent_list_n, a.tile_list_cap, std::max(a.n_xcc, 1));
        };
        e->last_raster_path = k2q ? (mesh ? MW_PATH_QUAD_MESH : MW_PATH_QUAD) : MW_PATH_TILE;
        if (mesh) {
            // the first part — every tile no mesh can touch: it needs nothing of the mesh kernels — on the quad stream beside them
            // (forked above, behind the geometry kernel); the mesh tiles end the chain on the caller's stream
            if (k2q) launch_k2q(1 << 4, e->quad_stream); else launch_k2(1 << 4, e->quad_stream);
            launch_k2(2 << 4, st);
            HIP_TRY(e, hipEventRecord(e->ev_mesh_join, e->quad_stream));
            HIP_TRY(e, hipStreamWaitEvent(st, e->ev_mesh_join, 0));
        } else if (k2q) {
            launch_k2q(0, st);
        } else {
            launch_k2(0, st);
        }
        if (mesh) e->mesh_keys_dirty = false;
    }
    if (timed) {
        (void)hipEventRecord(ev.c, st);
        e->ev_used.push_back(ev);
    }
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

}  // namespace

extern "C" {

const char *mw_last_error(const mw_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int mw_create(const mw_config *cfg, mw_engine **out)
{
    if (!cfg || !out) return fail(nullptr, MW_E_INVALID, "null argument");
    if (cfg->abi_version != MW_ABI_VERSION) return fail(nullptr, MW_E_INVALID, "ABI version mismatch: header %d, caller %d", MW_ABI_VERSION, cfg->abi_version);
    if (cfg->num_envs <= 0 || cfg->max_ents < 0 || cfg->max_polys <= 0 || cfg->max_segs <= 0 || cfg->max_visible <= 0)
        return fail(nullptr, MW_E_INVALID, "bad capacities");
    if (cfg->rng_mode != MW_RNG_PHILOX && cfg->rng_mode != MW_RNG_PCG64) return fail(nullptr, MW_E_INVALID, "unknown rng_mode %d", cfg->rng_mode);
    if (cfg->rng_mode == MW_RNG_PCG64 && cfg->generator == MW_GEN_NONE)
        return fail(nullptr, MW_E_INVALID, "MW_RNG_PCG64 (the reference's own numpy stream) needs a device generator");
    if (cfg->max_ents > 64) return fail(nullptr, MW_E_CAPACITY, "max_ents > 64 (one entity slot per lane of the env's wavefront)");
    if (cfg->msaa != 8 && cfg->msaa != 4 && cfg->msaa != 1) return fail(nullptr, MW_E_INVALID, "msaa must be 8, 4 or 1");
    if (cfg->obs_width % MW_TILE_W || cfg->obs_height % MW_TILE_H || cfg->obs_width > 255 * MW_TILE_W || cfg->obs_height > 255 * MW_TILE_H)
        return fail(nullptr, MW_E_INVALID, "obs size must be a multiple of %dx%d", MW_TILE_W, MW_TILE_H);
    if (cfg->max_visible > 60000) return fail(nullptr, MW_E_CAPACITY, "max_visible too large (16-bit draw ids)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, MW_E_DEVICE, "no HIP device available");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, MW_E_DEVICE, "device %d out of range (%d devices)", cfg->device_id, ndev);
    hipError_t st = hipSetDevice(cfg->device_id);
    if (st != hipSuccess) return fail(nullptr, MW_E_HIP, "hipSetDevice: %s", hipGetErrorString(st));

    mw_engine *e = new mw_engine();
    e->cfg = *cfg;
    const int N = cfg->num_envs, E = std::max(cfg->max_ents, 1);
    e->cfg.max_ents = E;
    e->n_sets = cfg->shared_geometry ? 1 : N;
    MwArgs &a = e->args;
    a.N = N; a.W = cfg->obs_width; a.H = cfg->obs_height; a.E = E;
    a.max_polys = cfg->max_polys; a.max_segs = cfg->max_segs;
    // triangle records per env: a polygon or box face is two triangles, clipping adds a few
    a.max_vis = cfg->max_visible * 6 < 60000 ? cfg->max_visible * 6 : 60000;
    a.shared_geom = cfg->shared_geometry ? 1 : 0;
    a.task = cfg->task; a.goal_ent = cfg->goal_ent; a.goal_ent2 = cfg->goal_ent2; a.num_objs = cfg->num_objs; a.max_steps = cfg->max_episode_steps;
    a.rng_mode = cfg->rng_mode;
    a.occlusion = 1;
    if (const char *s = getenv("MW_OCCLUSION")) a.occlusion = atoi(s) != 0;
    a.domain_rand = cfg->domain_rand; a.generator = cfg->generator; a.autoreset = cfg->autoreset;
    a.tiles_x = a.W / MW_TILE_W; a.tiles_y = a.H / MW_TILE_H; a.n_tiles = a.tiles_x * a.tiles_y;
    a.agent_radius = cfg->agent_radius; a.max_forward_step = cfg->max_forward_step;
    a.agent_height = cfg->agent_height > 0.0 ? cfg->agent_height : 1.6;
    a.fwd = cfg->forward_step; a.drift = cfg->forward_drift; a.turn = cfg->turn_step;
    memcpy(a.gen_args, cfg->gen_args, sizeof a.gen_args);
    MwGenTables gt{};
    memcpy(gt.gen_tab, cfg->gen_tab, sizeof gt.gen_tab);
    memcpy(gt.gen_colors, cfg->gen_colors, sizeof gt.gen_colors);
    memcpy(gt.tex_nvar, cfg->tex_nvar, sizeof gt.tex_nvar);
    memcpy(gt.tex_var_id, cfg->tex_var_id, sizeof gt.tex_var_id);
    memcpy(gt.tex_var_scale, cfg->tex_var_scale, sizeof gt.tex_var_scale);
    gt.room_wall_height = cfg->room_wall_height; gt.room_no_ceiling = cfg->room_no_ceiling;
    for (int i = 0; i < 3; ++i) {
        a.sky[i] = cfg->sky_color[i]; a.light_pos[i] = cfg->light_pos[i]; a.light_color[i] = cfg->light_color[i];
        a.light_ambient[i] = cfg->light_ambient[i]; a.color_bias[i] = cfg->obj_color_bias[i];
    }
    a.cam_height = cfg->cam_height; a.cam_fwd_disp = cfg->cam_fwd_disp; a.cam_pitch = cfg->cam_pitch; a.cam_fov_y = cfg->cam_fov_y;
    int rc = MW_OK;
#define ALLOC(ptr, count) if (rc == MW_OK) rc = dev_alloc(e, &ptr, (size_t)(count))
    ALLOC(a.ax, N); ALLOC(a.ay, N); ALLOC(a.az, N); ALLOC(a.adir, N);
    ALLOC(a.cam, 4 * (size_t)N); ALLOC(a.light, 12 * (size_t)N);
    ALLOC(a.carry, N); ALLOC(a.step, N); ALLOC(a.picked, N);
    if (cfg->task == MW_TASK_COLLECT) { ALLOC(a.health, N); ALLOC(a.final_health, N); }
    ALLOC(a.final_goal, 3 * (size_t)N);
    ALLOC(a.ekind, (size_t)E * N); ALLOC(a.emesh, (size_t)E * N); ALLOC(a.estatic, (size_t)E * N);
    ALLOC(a.epos, 3 * (size_t)E * N); ALLOC(a.edir, (size_t)E * N); ALLOC(a.egeom, 9 * (size_t)E * N);
    ALLOC(a.rng, 5 * (size_t)N); ALLOC(a.extent, 4 * (size_t)N);
    MwGenTables *d_gt = nullptr;
    ALLOC(d_gt, 1);
    if (rc == MW_OK) { (void)hipMemcpy(d_gt, &gt, sizeof gt, hipMemcpyHostToDevice); a.gt = d_gt; }
    mw_poly *polys = nullptr; int32_t *npolys = nullptr; double *segs = nullptr; int32_t *nsegs = nullptr;
    ALLOC(polys, (size_t)e->n_sets * cfg->max_polys); ALLOC(npolys, e->n_sets);
    ALLOC(segs, (size_t)e->n_sets * cfg->max_segs * 4); ALLOC(nsegs, e->n_sets);
    a.polys = polys; a.npolys = npolys; a.segs = segs; a.nsegs = nsegs;
    // spare mode: a pre-generated next world per env (mw_device.h::MwSpare)
    MwSpare sp{};
    // Spare worlds pay where the inline generator is the launch's tail: in the dense K1 of small scenes (a wave with a
    // finished env takes 26 us instead of 12, profiles/r02a) and in the Maze (a block regenerating its maze takes 300 us,
    // the launch with it; its refills run on the side stream, launch_frame); in the other wave-per-env scenes they bought
    // 2 us of 51 (Hallway) and stay off.  MW_SPARE=1 / 0 forces either.  Not with domain randomisation: the per-step draws
    // interleave with the worlds in the env's stream.
    {
        const bool small_scene = cfg->max_visible <= 64 && cfg->max_polys + 6 * std::max(cfg->max_ents, 1) <= 32;      // = the dense K1 (k1_dense_lanes)
        bool want = small_scene || cfg->generator == MW_GEN_MAZE;
        if (const char *s = getenv("MW_SPARE")) want = atoi(s) != 0;
        e->spare_mode = cfg->generator != MW_GEN_NONE && !cfg->domain_rand && want && cfg->task != MW_TASK_COLLECT;     // CollectHealth's respawns draw from the stream mid-episode
    }
    if (e->spare_mode) {
        ALLOC(sp.ax, N); ALLOC(sp.ay, N); ALLOC(sp.az, N); ALLOC(sp.adir, N);
        ALLOC(sp.cam, 4 * (size_t)N); ALLOC(sp.light, 12 * (size_t)N); ALLOC(sp.extent, 4 * (size_t)N);
        ALLOC(sp.ekind, (size_t)E * N); ALLOC(sp.emesh, (size_t)E * N); ALLOC(sp.estatic, (size_t)E * N);
        ALLOC(sp.epos, 3 * (size_t)E * N); ALLOC(sp.edir, (size_t)E * N); ALLOC(sp.egeom, 9 * (size_t)E * N);
        if (!cfg->shared_geometry) {
            ALLOC(sp.polys, (size_t)e->n_sets * cfg->max_polys); ALLOC(sp.npolys, e->n_sets);
            ALLOC(sp.segs, (size_t)e->n_sets * cfg->max_segs * 4); ALLOC(sp.nsegs, e->n_sets);
        }
        MwSpare *d_sp = nullptr;
        ALLOC(d_sp, 1);
        ALLOC(a.refill_mask, N);
        ALLOC(e->d_spare_dummy, 3 * (size_t)N);
        e->spare_host = sp;
        if (rc == MW_OK) {      // every spare starts out consumed: the first reset generates it
            std::vector<uint32_t> ones((size_t)N, 1u);
            (void)hipMemcpy(a.refill_mask, ones.data(), 4 * (size_t)N, hipMemcpyHostToDevice);
        }
        if (rc == MW_OK) { (void)hipMemcpy(d_sp, &sp, sizeof sp, hipMemcpyHostToDevice); a.spare = d_sp; }
    }
    ALLOC(e->d_meshdesc, MW_MAX_MESH);
    a.mesh = e->d_meshdesc;      // a.tex / a.texels: upload_textures
    ALLOC(a.rec_raster, (size_t)N * a.max_vis * MW_RASTER_REC);
    ALLOC(a.rec_shade, (size_t)N * a.max_vis * MW_SHADE_REC);
    ALLOC(a.rec_cull, (size_t)N * a.max_vis * MW_CULL_REC);
    if (cfg->max_visible > 64) {
        // big scenes: the visiting order the geometry kernel leaves for K2 (mw_geom.hip)
        ALLOC(a.rec_order, (size_t)N * (a.max_vis + 1));
        if (rc == MW_OK) (void)hipMemset(a.rec_order, 0, (size_t)N * (a.max_vis + 1) * 2);
    }
    if (cfg->max_polys > 64 && !(getenv("MW_OCC_CACHE") && atoi(getenv("MW_OCC_CACHE")) == 0)) {
        // big scenes: the geometry kernel's per-world culling data (mw_geom.hip), zeroed = nothing cached
        ALLOC(a.occ_valid, e->n_sets);
        ALLOC(a.occ_cache, (size_t)e->n_sets * MW_OCC_CACHE_STRIDE(cfg->max_polys));
    }
    ALLOC(a.pending_remove, (size_t)N);
    if (rc == MW_OK) (void)hipMemset(a.pending_remove, 0xFF, 4 * (size_t)N);
    ALLOC(a.nvis, N); ALLOC(a.envhdr, (size_t)MW_ENVHDR * N); ALLOC(a.status, 1);
    ALLOC(e->d_reward_scratch, N); ALLOC(e->d_flag_scratch, 2 * (size_t)N); ALLOC(e->d_action_scratch, N);
    ALLOC(e->d_mask, N); ALLOC(e->d_step_override, 3 * (size_t)N);
#ifdef MW_PERF_HOOKS        // (tools/perf: make EXTRA=-DMW_PERF_HOOKS — kernel phase stamps dumped by mw_destroy; not in the product build)
    if (getenv("MW_K1_PROF")) {     // per-env cycle stamps of the geometry kernel's phases
        ALLOC(a.k1_prof, MW_K1_PROF_SLOTS * (size_t)N);
        if (rc == MW_OK) (void)hipMemset(a.k1_prof, 0, 8 * MW_K1_PROF_SLOTS * (size_t)N);
    }
#endif
#undef ALLOC
    if (rc != MW_OK) { g_create_error = e->err; mw_destroy(e); return rc; }
    // carrying = -1 everywhere; default seeds = env index
    {
        std::vector<int32_t> m1((size_t)N, -1);
        (void)hipMemcpy(a.carry, m1.data(), 4 * (size_t)N, hipMemcpyHostToDevice);
        std::vector<uint64_t> seeds(5 * (size_t)N, 0);
        for (int i = 0; i < N; ++i) seed_env(e, seeds.data(), i, (uint64_t)i);
        (void)hipMemcpy(a.rng, seeds.data(), 40 * (size_t)N, hipMemcpyHostToDevice);
    }
    e->mesh_desc.assign(MW_MAX_MESH, MwMeshDesc{});
    e->mesh_pos.assign(MW_MAX_MESH, {}); e->mesh_nrm.assign(MW_MAX_MESH, {}); e->mesh_rgb.assign(MW_MAX_MESH, {}); e->mesh_uv.assign(MW_MAX_MESH, {});
    e->mesh_vtab.assign(MW_MAX_MESH, {}); e->mesh_itab.assign(MW_MAX_MESH, {});
    e->tex_desc.assign(MW_MAX_TEX, MwTexDesc{});
    e->tex_data.assign(MW_MAX_TEX, {});
    if (upload_textures(e) != MW_OK) { g_create_error = e->err; mw_destroy(e); return MW_E_HIP; }
    e->waves_per_env = pick_waves_per_env(e);
    {
        // The XCDs of this device as workgroups see them (HW_REG_XCC_ID of 256 workgroups: 8, 4, 2 or one id per partition mode).
        // The mesh path files an env's entities and mesh tiles under class env % n_xcc and workgroup b of the entity / tile launches
        // draws from class b % n_xcc: where workgroup b runs on XCD b % n_xcc — every launch of a fresh process — an env's records,
        // keys and planes meet one L2 (the mesh tiles' fetch 50 -> 37 MB).  Locality only: nothing is wrong when the dispatcher's
        // round-robin starts elsewhere.
        uint32_t *d_ids = nullptr, ids[256];
        a.n_xcc = 0;
        if (hipMalloc((void **)&d_ids, sizeof ids) == hipSuccess) {
            hipLaunchKernelGGL(mw_xcc_probe_kernel, dim3(256), dim3(64), 0, 0, d_ids);
            if (hipMemcpy(ids, d_ids, sizeof ids, hipMemcpyDeviceToHost) == hipSuccess) {
                uint32_t seen = 0u;
                for (uint32_t v : ids) seen |= 1u << (v & 15u);
                for (int n : {8, 4, 2}) if (seen == (1u << n) - 1u) a.n_xcc = n;
            }
            (void)hipFree(d_ids);
        }
        if (a.n_xcc == 0) a.n_xcc = 1;      // (one id, or a set this code does not know: one class — the lists are about locality only)
    }
    if (const char *s = getenv("MW_DEBUG_FLAGS")) e->dbg_flags = atoi(s);
    if (const char *s = getenv("MW_K2Q")) e->use_k2q = atoi(s) != 0;
    if (const char *s = getenv("MW_GENERIC_RASTER")) e->generic_raster = atoi(s) != 0;
#ifdef MW_PERF_HOOKS
    if (getenv("MW_ENT_PROF")) { if (dev_alloc(e, &e->d_ent_prof, (size_t)16 * 2 * N * MW_MAX_MESH_ENTS * 8) != MW_OK) { g_create_error = e->err; mw_destroy(e); return MW_E_NOMEM; } }
    if (getenv("MW_K2Q_PROF")) { if (dev_alloc(e, &e->d_k2q_prof, (size_t)N * 80) != MW_OK) { g_create_error = e->err; mw_destroy(e); return MW_E_NOMEM; } }
#endif
    {
        // the quad kernel (mw_rasterq.hip) keeps an env's frame, quad lists and triangle records in LDS: frames up to 8192 pixels
        const int S = cfg->msaa == 4 ? 4 : 8;
        const int lds = mw_rasterq_lds_bytes(S, a.W, a.H, a.n_tiles, 1);
        e->k2q_ok = (cfg->msaa == 8 || cfg->msaa == 4) && a.W <= 128 && a.H <= 128 && a.W * a.H <= 8192 && lds <= 64 * 1024;
    }
    if (sync_gen_args(e) != MW_OK) { g_create_error = e->err; mw_destroy(e); return MW_E_HIP; }
    *out = e;
    return MW_OK;
}

void mw_destroy(mw_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
#ifdef MW_PERF_HOOKS
    if (e->d_k2q_prof) {
        std::vector<unsigned long long> h((size_t)e->cfg.num_envs * 80);
        if (hipMemcpy(h.data(), e->d_k2q_prof, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen(getenv("MW_K2Q_PROF"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    if (e->d_ent_prof) {
        std::vector<unsigned long long> h((size_t)16 * 2 * e->cfg.num_envs * MW_MAX_MESH_ENTS * 8);
        if (hipMemcpy(h.data(), e->d_ent_prof, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen(getenv("MW_ENT_PROF"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    if (e->args.k1_prof) {
        std::vector<unsigned long long> h((size_t)e->cfg.num_envs * MW_K1_PROF_SLOTS);
        if (hipMemcpy(h.data(), e->args.k1_prof, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen(getenv("MW_K1_PROF"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    if (getenv("MW_SLOW_STATS") && e->d_slow_count) {      // perf experiments only: the last frame's slow fragments per env
        std::vector<int32_t> h((size_t)e->cfg.num_envs * 4);
        if (hipMemcpy(h.data(), e->d_slow_count, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            long long tot = 0, nz = 0, mx = 0;
            for (int i = 0; i < e->cfg.num_envs; ++i) { const int v = h[(size_t)e->cfg.num_envs + i] + h[(size_t)e->cfg.num_envs * 3 + i]; tot += v; nz += v > 0; mx = std::max<long long>(mx, v); }
            fprintf(stderr, "slow fragments: total %lld, envs with any %lld of %d, max %lld\n", tot, nz, e->cfg.num_envs, mx);
            int32_t c[2 * MW_CNT_WORDS];
            if (e->d_ent_counter && hipMemcpy(c, e->d_ent_counter, sizeof c, hipMemcpyDeviceToHost) == hipSuccess)
                for (int p = 0; p < 2; ++p) {
                    int nl = 0, ns = 0, nt = 0;
                    for (int x = 0; x < 8; ++x) { nl += c[p * MW_CNT_WORDS + MW_CNT_LONG + x]; ns += c[p * MW_CNT_WORDS + MW_CNT_SHORT + x]; nt += c[p * MW_CNT_WORDS + MW_CNT_TILES + x]; }
                    fprintf(stderr, "work lists (parity %d): long meshes %d, short %d, mesh tiles %d, envs with slow-path triangles %d\n", p, nl, ns,
                            nt, c[p * MW_CNT_WORDS + MW_CNT_SLOW_ENVS]);
                }
        }
    }
#endif
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->d_texels) (void)hipFree(e->d_texels);
    for (float *p : {e->d_mesh_pos, e->d_mesh_nrm, e->d_mesh_rgb, e->d_mesh_uv, e->d_mesh_stream, e->d_mesh_attr}) if (p) (void)hipFree(p);
    if (e->d_mesh_vpos) (void)hipFree(e->d_mesh_vpos);
    if (e->d_mesh_idx) (void)hipFree(e->d_mesh_idx);
    if (e->d_view_keys) (void)hipFree(e->d_view_keys);
    if (e->d_plane_cache) (void)hipFree(e->d_plane_cache);
    if (e->d_mesh_keys) (void)hipFree(e->d_mesh_keys);
    if (e->d_ent_list) (void)hipFree(e->d_ent_list);
    if (e->d_tile_list) (void)hipFree(e->d_tile_list);
    if (e->d_slow_envs) (void)hipFree(e->d_slow_envs);
    for (void *q : {(void *)e->d_slow_count, (void *)e->d_slow_tris, (void *)e->d_slow_frags, (void *)e->d_slow_head}) if (q) (void)hipFree(q);
    if (e->quad_stream) { (void)hipStreamDestroy(e->quad_stream); (void)hipEventDestroy(e->ev_mesh_fork); (void)hipEventDestroy(e->ev_mesh_join); }
    if (e->side_stream) { (void)hipStreamDestroy(e->side_stream); (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join); }
    for (auto &ev : e->ev_used) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); (void)hipEventDestroy(ev.c); }
    for (auto &ev : e->ev_free) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); (void)hipEventDestroy(ev.c); }
    delete e;
}

int mw_upload_texture(mw_engine *e, int32_t tex_id, const uint8_t *rgb, int32_t w, int32_t h)
{
    if (!e || !rgb) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE_SYNC(e);
    if (tex_id < 0 || tex_id >= MW_MAX_TEX) return fail(e, MW_E_CAPACITY, "texture id %d out of range (max %d)", tex_id, MW_MAX_TEX);
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384) return fail(e, MW_E_INVALID, "bad texture size %dx%d", w, h);
    build_pyramid(rgb, w, h, e->tex_data[tex_id], e->tex_desc[tex_id]);
    return upload_textures(e);
}

int mw_upload_mesh(mw_engine *e, int32_t mesh_id, const float *pos, const float *nrm, const float *uv,
                   const float *rgb, int32_t ntris, int32_t tex_id)
{
    if (!e || !pos || !nrm || !rgb) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE_SYNC(e);
    if (tex_id >= MW_MAX_TEX || (tex_id >= 0 && !uv)) return fail(e, MW_E_INVALID, "textured mesh needs texcoords and a valid texture id");
    if (mesh_id < 0 || mesh_id >= MW_MAX_MESH) return fail(e, MW_E_CAPACITY, "mesh id %d out of range (max %d)", mesh_id, MW_MAX_MESH);
    if (ntris <= 0 || ntris > 60000) return fail(e, MW_E_CAPACITY, "mesh with %d triangles (1..60000 supported: 16-bit draw ids)", ntris);
    // storage order: triangles sorted by the direction of their face normal (octahedral map, 6 + 6 bit Morton code,
    // stable), mw_device.h: MW_MESH_POS_STRIDE
    std::vector<uint32_t> order((size_t)ntris), key((size_t)ntris);
    for (int t = 0; t < ntris; ++t) {
        const float *p = pos + (size_t)t * 9;
        const double ax = p[3] - p[0], ay = p[4] - p[1], az = p[5] - p[2], bx = p[6] - p[0], by = p[7] - p[1], bz = p[8] - p[2];
        double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
        double l1 = std::fabs(nx) + std::fabs(ny) + std::fabs(nz);
        if (!(l1 > 0.0)) { nx = nrm[(size_t)t * 9]; ny = nrm[(size_t)t * 9 + 1]; nz = nrm[(size_t)t * 9 + 2]; l1 = std::fabs(nx) + std::fabs(ny) + std::fabs(nz); }
        if (!(l1 > 0.0)) { nx = 0; ny = 1; nz = 0; l1 = 1; }
        double u = nx / l1, v = nz / l1;
        if (ny < 0.0) {     // lower hemisphere folded outwards
            const double uu = (1.0 - std::fabs(v)) * (u >= 0 ? 1.0 : -1.0), vv = (1.0 - std::fabs(u)) * (v >= 0 ? 1.0 : -1.0);
            u = uu; v = vv;
        }
        const uint32_t qu = (uint32_t)std::min(63.0, std::max(0.0, (u * 0.5 + 0.5) * 64.0)), qv = (uint32_t)std::min(63.0, std::max(0.0, (v * 0.5 + 0.5) * 64.0));
        uint32_t m = 0;
        for (int b = 0; b < 6; ++b) m |= ((qu >> b) & 1u) << (2 * b) | ((qv >> b) & 1u) << (2 * b + 1);
        key[t] = m; order[t] = (uint32_t)t;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    auto &P = e->mesh_pos[mesh_id]; auto &Nn = e->mesh_nrm[mesh_id]; auto &Cc = e->mesh_rgb[mesh_id]; auto &U = e->mesh_uv[mesh_id];
    P.assign((size_t)ntris * MW_MESH_POS_STRIDE, 0.0f);
    Nn.assign(nrm, nrm + (size_t)ntris * 9); Cc.assign(rgb, rgb + (size_t)ntris * 9);
    if (uv) U.assign(uv, uv + (size_t)ntris * 6); else U.assign((size_t)ntris * 6, 0.0f);
    for (int i = 0; i < ntris; ++i) {
        memcpy(&P[(size_t)i * MW_MESH_POS_STRIDE], pos + (size_t)i * 9, 36);
        memcpy(&P[(size_t)i * MW_MESH_POS_STRIDE + 9], &order[i], 4);      // the i-th triangle of the rasterisation order
    }
    {
        // the table of distinct positions (bit patterns: -0 and 0 stay apart) and the triangles' indices into it, in
        // rasterisation order; a mesh with more than MW_MESH_VCAP positions keeps none (the entity kernel then takes its
        // triangles through the vertex stage one by one)
        std::map<std::array<uint32_t, 3>, uint32_t> seen;
        auto &VT = e->mesh_vtab[mesh_id]; auto &IT = e->mesh_itab[mesh_id];
        VT.clear(); IT.assign((size_t)ntris * 2, 0u);
        bool fits = true;
        for (int k = 0; k < ntris && fits; ++k) {
            const uint32_t tri = order[k];
            uint32_t ix[3];
            for (int c = 0; c < 3; ++c) {
                std::array<uint32_t, 3> key;
                memcpy(key.data(), pos + ((size_t)tri * 3 + c) * 3, 12);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    if (seen.size() >= MW_MESH_VCAP) { fits = false; break; }
                    it = seen.emplace(key, (uint32_t)seen.size()).first;
                    const float *pp = pos + ((size_t)tri * 3 + c) * 3;
                    VT.insert(VT.end(), {pp[0], pp[1], pp[2], 0.0f});
                }
                ix[c] = it->second;
            }
            IT[(size_t)k * 2] = ix[0] | (ix[1] << 16);
            IT[(size_t)k * 2 + 1] = ix[2] | (tri << 16);
        }
        if (!fits) VT.clear();
        e->mesh_desc[mesh_id].nverts = (uint32_t)(VT.size() / 4);
    }
    memcpy(e->mesh_desc[mesh_id].last_n, nrm + ((size_t)(ntris - 1) * 3 + 2) * 3, 12);
    e->mesh_desc[mesh_id].ntris = (uint32_t)ntris;
    e->mesh_desc[mesh_id].tex = tex_id;
    {
        float r2 = 0.0f;
        for (size_t i = 0; i < (size_t)ntris * 3; ++i)
            r2 = std::max(r2, pos[i * 3] * pos[i * 3] + pos[i * 3 + 1] * pos[i * 3 + 1] + pos[i * 3 + 2] * pos[i * 3 + 2]);
        const float r = std::sqrt(r2) * 1.0001f;
        memcpy(&e->mesh_desc[mesh_id].bound_bits, &r, 4);
        // bounding box, its centre, the sphere about the centre (doubles: the radius rounds up)
        MwMeshDesc &md = e->mesh_desc[mesh_id];
        for (int c = 0; c < 3; ++c) { md.bmin[c] = pos[c]; md.bmax[c] = pos[c]; }
        for (size_t i = 0; i < (size_t)ntris * 3; ++i)
            for (int c = 0; c < 3; ++c) { md.bmin[c] = std::min(md.bmin[c], pos[i * 3 + c]); md.bmax[c] = std::max(md.bmax[c], pos[i * 3 + c]); }
        for (int c = 0; c < 3; ++c) md.center[c] = 0.5f * (md.bmin[c] + md.bmax[c]);
        double rc2 = 0.0;
        for (size_t i = 0; i < (size_t)ntris * 3; ++i) {
            double d2 = 0.0;
            for (int c = 0; c < 3; ++c) { const double d = (double)pos[i * 3 + c] - (double)md.center[c]; d2 += d * d; }
            rc2 = std::max(rc2, d2);
        }
        md.radius = (float)(std::sqrt(rc2) * 1.0001 + 1e-6);
    }
    // repack all pools (uploads are rare)
    size_t total = 0;
    size_t total_v = 0;
    e->max_mesh_verts = 0;
    for (int i = 0; i < MW_MAX_MESH; ++i) {
        e->mesh_desc[i].first = (uint32_t)total; total += e->mesh_desc[i].ntris;
        e->mesh_desc[i].vfirst = (uint32_t)total_v; total_v += e->mesh_desc[i].nverts;
        e->max_mesh_verts = std::max(e->max_mesh_verts, (int)e->mesh_desc[i].nverts);
    }
    if (e->d_mesh_vpos) { (void)hipFree(e->d_mesh_vpos); e->d_mesh_vpos = nullptr; }
    if (e->d_mesh_idx) { (void)hipFree(e->d_mesh_idx); e->d_mesh_idx = nullptr; }
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_vpos, std::max<size_t>(total_v, 1) * 16));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_idx, total * 8));
    for (float **p : {&e->d_mesh_pos, &e->d_mesh_nrm, &e->d_mesh_rgb, &e->d_mesh_uv, &e->d_mesh_stream, &e->d_mesh_attr})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_pos, total * 4 * MW_MESH_POS_STRIDE));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_nrm, total * 36));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_rgb, total * 36));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_uv, total * 24));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_stream, total * 48));
    HIP_TRY(e, hipMalloc((void **)&e->d_mesh_attr, total * 96));
    for (int i = 0; i < MW_MAX_MESH; ++i) {
        const size_t n = e->mesh_desc[i].ntris, off = (size_t)e->mesh_desc[i].first * 9;
        if (!n) continue;
        HIP_TRY(e, hipMemcpy(e->d_mesh_pos + (size_t)e->mesh_desc[i].first * MW_MESH_POS_STRIDE, e->mesh_pos[i].data(), n * 4 * MW_MESH_POS_STRIDE, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->d_mesh_nrm + off, e->mesh_nrm[i].data(), n * 36, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->d_mesh_rgb + off, e->mesh_rgb[i].data(), n * 36, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->d_mesh_uv + (size_t)e->mesh_desc[i].first * 6, e->mesh_uv[i].data(), n * 24, hipMemcpyHostToDevice));
        if (e->mesh_desc[i].nverts)
            HIP_TRY(e, hipMemcpy(e->d_mesh_vpos + e->mesh_desc[i].vfirst, e->mesh_vtab[i].data(), (size_t)e->mesh_desc[i].nverts * 16, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->d_mesh_idx + e->mesh_desc[i].first, e->mesh_itab[i].data(), n * 8, hipMemcpyHostToDevice));
        {
            // the scatter kernel's stream: the triangles in rasterisation order, 48 bytes each (9 coordinates, the triangle's index)
            std::vector<float> st(n * 12, 0.0f);
            const auto &P = e->mesh_pos[i];
            for (size_t k = 0; k < n; ++k) {
                uint32_t tri;
                memcpy(&tri, &P[k * MW_MESH_POS_STRIDE + 9], 4);
                memcpy(&st[k * 12], &P[(size_t)tri * MW_MESH_POS_STRIDE], 36);
                memcpy(&st[k * 12 + 9], &tri, 4);
            }
            HIP_TRY(e, hipMemcpy(e->d_mesh_stream + (size_t)e->mesh_desc[i].first * 12, st.data(), n * 48, hipMemcpyHostToDevice));
            // ... and their vertex attributes in the same order, 96 bytes each (normals, colours, texture coordinates)
            std::vector<float> at(n * 24, 0.0f);
            for (size_t k = 0; k < n; ++k) {
                uint32_t tri;
                memcpy(&tri, &P[k * MW_MESH_POS_STRIDE + 9], 4);
                memcpy(&at[k * 24], &e->mesh_nrm[i][(size_t)tri * 9], 36);
                memcpy(&at[k * 24 + 9], &e->mesh_rgb[i][(size_t)tri * 9], 36);
                memcpy(&at[k * 24 + 18], &e->mesh_uv[i][(size_t)tri * 6], 24);
            }
            HIP_TRY(e, hipMemcpy(e->d_mesh_attr + (size_t)e->mesh_desc[i].first * 24, at.data(), n * 96, hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(e, hipMemcpy(e->d_meshdesc, e->mesh_desc.data(), sizeof(MwMeshDesc) * MW_MAX_MESH, hipMemcpyHostToDevice));
    e->args.mesh_pos = e->d_mesh_pos; e->args.mesh_nrm = e->d_mesh_nrm; e->args.mesh_rgb = e->d_mesh_rgb; e->args.mesh_uv = e->d_mesh_uv;
    if (e->d_gen_live && sync_gen_args(e) != MW_OK) return MW_E_HIP;
    e->have_meshes = true;
    e->max_mesh_tris = std::max(e->max_mesh_tris, (int)ntris);
    return ensure_mesh_buffers(e);
}

int mw_set_geometry(mw_engine *e, int32_t env, const mw_poly *polys, int32_t n_polys, const double *segs, int32_t n_segs)
{
    if (!e || (n_polys > 0 && !polys) || (n_segs > 0 && !segs)) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE_SYNC(e);
    if (n_polys < 0 || n_polys > e->cfg.max_polys) return fail(e, MW_E_CAPACITY, "%d polygons > max_polys %d", n_polys, e->cfg.max_polys);
    if (n_segs < 0 || n_segs > e->cfg.max_segs) return fail(e, MW_E_CAPACITY, "%d segments > max_segs %d", n_segs, e->cfg.max_segs);
    int set = 0;
    if (e->cfg.shared_geometry) {
        if (env != -1) return fail(e, MW_E_INVALID, "engine uses one shared geometry set: pass env = -1");
    } else {
        if (env < 0 || env >= e->cfg.num_envs) return fail(e, MW_E_INVALID, "env %d out of range", env);
        set = env;
    }
    for (int i = 0; i < n_polys; ++i) {
        const int nv = polys[i].nv & 0xFF;
        if (nv != 3 && nv != 4) return fail(e, MW_E_INVALID, "polygon %d has %d vertices (3 or 4 supported)", i, nv);
        if (polys[i].tex >= MW_MAX_TEX) return fail(e, MW_E_INVALID, "polygon %d: bad texture id", i);
        if (polys[i].tex >= 0 && e->tex_desc[polys[i].tex].nlevels == 0) return fail(e, MW_E_INVALID, "polygon %d uses texture %d which was never uploaded", i, polys[i].tex);
    }
    HIP_TRY(e, hipMemcpy(const_cast<mw_poly *>(e->args.polys) + (size_t)set * e->cfg.max_polys, polys, sizeof(mw_poly) * (size_t)n_polys, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(const_cast<int32_t *>(e->args.npolys) + set, &n_polys, 4, hipMemcpyHostToDevice));
    if (e->args.occ_valid) HIP_TRY(e, hipMemset(e->args.occ_valid + set, 0, 4));
    HIP_TRY(e, hipMemcpy(const_cast<double *>(e->args.segs) + (size_t)set * e->cfg.max_segs * 4, segs, 32 * (size_t)n_segs, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(const_cast<int32_t *>(e->args.nsegs) + set, &n_segs, 4, hipMemcpyHostToDevice));
    return MW_OK;
}

int mw_get_geometry(mw_engine *e, int32_t env, mw_poly *polys, int32_t *n_polys, double *segs, int32_t *n_segs)
{
    if (!e || !polys || !n_polys || !segs || !n_segs) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE_SYNC(e);
    const int set = e->cfg.shared_geometry ? 0 : env;
    if (set < 0 || set >= e->n_sets) return fail(e, MW_E_INVALID, "env out of range");
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(n_polys, e->args.npolys + set, 4, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(n_segs, e->args.nsegs + set, 4, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(polys, e->args.polys + (size_t)set * e->cfg.max_polys, sizeof(mw_poly) * (size_t)e->cfg.max_polys, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(segs, e->args.segs + (size_t)set * e->cfg.max_segs * 4, 32 * (size_t)e->cfg.max_segs, hipMemcpyDeviceToHost));
    return MW_OK;
}

int mw_set_state(mw_engine *e, int32_t first_env, int32_t count, const mw_state_view *host)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE_SYNC(e);
    return state_xfer(e, first_env, count, host, true);
}

int mw_get_state(mw_engine *e, int32_t first_env, int32_t count, mw_state_view *host)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE_SYNC(e);
    (void)hipDeviceSynchronize();
    return state_xfer(e, first_env, count, host, false);
}

int mw_set_gen_program(mw_engine *e, const mw_gen_program *prog, const mw_poly *polys, const int32_t *poly_room,
                       const int32_t *poly_surf, const double *poly_m, int32_t n_polys, const double *segs, int32_t n_segs)
{
    if (!e || !prog) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE_SYNC(e);
    if (prog->n_rooms < 1 || prog->n_rooms > MW_PROG_MAX_ROOMS || prog->n_tex < 0 || prog->n_tex > MW_PROG_MAX_TEX ||
        prog->n_ops < 0 || prog->n_ops > MW_PROG_MAX_OPS || prog->n_ents < 0 || prog->n_ents > MW_PROG_MAX_ENTS ||
        prog->n_ents > e->cfg.max_ents || prog->sign_n < 0 || prog->sign_n > 8)
        return fail(e, MW_E_CAPACITY, "placement program exceeds the table sizes");
    if (n_polys < 0 || n_polys > e->cfg.max_polys || n_segs < 0 || n_segs > e->cfg.max_segs)
        return fail(e, MW_E_CAPACITY, "template geometry exceeds max_polys / max_segs");
    if (n_polys > 0 && (!polys || !poly_room || !poly_surf || !poly_m)) return fail(e, MW_E_INVALID, "null template geometry");
    for (int i = 0; i < prog->n_ops; ++i) {
        const mw_prog_op &op = prog->ops[i];
        const bool needs_slot = op.op == MW_OP_PLACE || op.op == MW_OP_FIXED || op.op == MW_OP_BOX_SIZE || op.op == MW_OP_COLOR || op.op == MW_OP_APPEND;
        if (op.op < MW_OP_COIN || op.op > MW_OP_APPEND) return fail(e, MW_E_INVALID, "op %d: unknown opcode %d", i, op.op);
        if (needs_slot && (op.slot >= prog->n_ents || (op.slot < 0 && !(op.op == MW_OP_PLACE || op.op == MW_OP_FIXED))))
            return fail(e, MW_E_INVALID, "op %d: bad entity slot %d", i, op.slot);
        if (op.op == MW_OP_PLACE && op.room >= prog->n_rooms) return fail(e, MW_E_INVALID, "op %d: bad room %d", i, op.room);
    }
    MwProgram hp{};
    hp.p = *prog;
    hp.n_polys = n_polys; hp.n_segs = n_segs;
    mw_poly *d_polys = nullptr; int32_t *d_room = nullptr, *d_surf = nullptr; double *d_m = nullptr, *d_segs = nullptr;
    int rc = MW_OK;
    if (rc == MW_OK) rc = dev_alloc(e, &d_polys, (size_t)n_polys);
    if (rc == MW_OK) rc = dev_alloc(e, &d_room, (size_t)n_polys);
    if (rc == MW_OK) rc = dev_alloc(e, &d_surf, (size_t)n_polys);
    if (rc == MW_OK) rc = dev_alloc(e, &d_m, (size_t)n_polys * 8);
    if (rc == MW_OK) rc = dev_alloc(e, &d_segs, (size_t)n_segs * 4);
    if (rc == MW_OK && !e->d_prog) rc = dev_alloc(e, &e->d_prog, 1);
    if (rc != MW_OK) return rc;
    if (n_polys > 0) {
        HIP_TRY(e, hipMemcpy(d_polys, polys, sizeof(mw_poly) * (size_t)n_polys, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(d_room, poly_room, 4 * (size_t)n_polys, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(d_surf, poly_surf, 4 * (size_t)n_polys, hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(d_m, poly_m, 64 * (size_t)n_polys, hipMemcpyHostToDevice));
    }
    if (n_segs > 0) HIP_TRY(e, hipMemcpy(d_segs, segs, 32 * (size_t)n_segs, hipMemcpyHostToDevice));
    hp.polys = d_polys; hp.poly_room = d_room; hp.poly_surf = d_surf; hp.poly_m = d_m; hp.segs = d_segs;
    HIP_TRY(e, hipMemcpy(e->d_prog, &hp, sizeof hp, hipMemcpyHostToDevice));
    e->args.prog = e->d_prog;
    if (e->cfg.shared_geometry && n_polys > 0) {        // no texture randomisation: the template IS the geometry
        const int r2 = mw_set_geometry(e, -1, polys, n_polys, segs, n_segs);
        if (r2 != MW_OK) return r2;
    }
    return sync_gen_args(e);
}

int mw_set_step_params(mw_engine *e, const double *host_params)
{
    if (!e) return MW_E_INVALID;
    if (!host_params) { e->use_step_override = false; return MW_OK; }
    ON_DEVICE_SYNC(e);
    HIP_TRY(e, hipMemcpy(e->d_step_override, host_params, 24 * (size_t)e->cfg.num_envs, hipMemcpyHostToDevice));
    e->use_step_override = true;
    return MW_OK;
}

int mw_reset(mw_engine *e, const uint8_t *mask, const uint64_t *seeds, void *stream)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE_SYNC(e);
    if (e->cfg.generator == MW_GEN_NONE && !seeds) return fail(e, MW_E_INVALID, "engine was created without a device-side generator");
    if (e->cfg.generator == MW_GEN_PROGRAM && !e->args.prog) return fail(e, MW_E_INVALID, "MW_GEN_PROGRAM: no placement program installed (mw_set_gen_program)");
    const int N = e->cfg.num_envs;
    hipStream_t st = (hipStream_t)stream;
    if (seeds) {
        std::vector<uint64_t> cur(5 * (size_t)N);
        HIP_TRY(e, hipStreamSynchronize(st));
        HIP_TRY(e, hipMemcpy(cur.data(), e->args.rng, 40 * (size_t)N, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            if (!mask || mask[i]) seed_env(e, cur.data(), i, seeds[i]);
        HIP_TRY(e, hipMemcpy(e->args.rng, cur.data(), 40 * (size_t)N, hipMemcpyHostToDevice));
    }
    // host-generated worlds (MW_GEN_NONE): seeds only re-seed the env's device stream, which then serves the per-step
    // domain-randomisation draws (miniworld.py:677-680); the world itself comes through mw_set_state / mw_set_geometry
    if (e->cfg.generator == MW_GEN_NONE) return MW_OK;
    if (mask) HIP_TRY(e, hipMemcpyAsync(e->d_mask, mask, N, hipMemcpyHostToDevice, st));
    const bool pcg = e->cfg.rng_mode == MW_RNG_PCG64;
    auto gen = pcg ? mw_reset_pcg_kernel : mw_reset_kernel;
    const dim3 grid(e->cfg.generator == MW_GEN_MAZE ? N : (N + 63) / 64);
    const int all = mask ? 0 : 1;
    if (!e->spare_mode) {
        hipLaunchKernelGGL(gen, grid, dim3(64), 0, st, e->args, (const uint8_t *)e->d_mask, all, 0);
    } else {
        // spare mode: a fresh seed generates the live world directly; without seeds the env's pre-generated world is
        // taken (what the same-step auto-reset does, after the pending refills have been run); either way the spare
        // of a reset env is then regenerated from its stream
        auto refill = pcg ? mw_refill_pcg_kernel : mw_refill_kernel;
        if (seeds) {
            hipLaunchKernelGGL(gen, grid, dim3(64), 0, st, e->args, (const uint8_t *)e->d_mask, all, 1);
        } else {
            hipLaunchKernelGGL(refill, grid, dim3(64), 0, st, e->args);
            hipLaunchKernelGGL(mw_take_spare_kernel, dim3(N), dim3(64), 0, st, e->args, (const uint8_t *)e->d_mask, all);
        }
        hipLaunchKernelGGL(refill, grid, dim3(64), 0, st, e->args);
    }
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

int mw_step(mw_engine *e, const int32_t *d_actions, uint8_t *d_obs, float *d_depth, float *d_reward,
            uint8_t *d_term, uint8_t *d_trunc, void *stream)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE(e);
    if (!d_actions) return fail(e, MW_E_INVALID, "d_actions is null");
    if ((e->cfg.generator == MW_GEN_PROGRAM || e->cfg.task >= MW_TASK_SIDEWALK) && !e->args.prog)
        return fail(e, MW_E_INVALID, "no placement program installed (mw_set_gen_program)");
    return launch_frame(e, true, 0, d_actions, d_obs, d_depth, d_reward, d_term, d_trunc, (hipStream_t)stream);
}

int mw_render(mw_engine *e, uint8_t *d_obs, float *d_depth, void *stream)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE(e);
    return launch_frame(e, false, 0, e->d_action_scratch, d_obs, d_depth, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int mw_render_top(mw_engine *e, uint8_t *d_obs, float *d_depth, int32_t render_agent, void *stream)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE(e);
    return launch_frame(e, false, 1 | (render_agent ? 2 : 0), e->d_action_scratch, d_obs, d_depth, nullptr, nullptr, nullptr,
                        (hipStream_t)stream);
}

int mw_render_view(mw_engine *e, int32_t env, int32_t view_flags, int32_t width, int32_t height, int32_t msaa,
                   uint8_t *d_out, float *d_depth, void *stream)
{
    if (!e || !d_out) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE(e);
    if (env < 0 || env >= e->cfg.num_envs) return fail(e, MW_E_INVALID, "env %d out of range", env);
    if (msaa != 1 && msaa != 4 && msaa != 8 && msaa != 16) return fail(e, MW_E_INVALID, "msaa must be 1, 4, 8 or 16");
    if (width <= 0 || height <= 0 || width % MW_TILE_W || height % MW_TILE_H || width > 255 * MW_TILE_W || height > 255 * MW_TILE_H)
        return fail(e, MW_E_INVALID, "frame buffer size must be a multiple of %dx%d", MW_TILE_W, MW_TILE_H);
    hipStream_t st = (hipStream_t)stream;
    MwArgs b = e->args;
    b.step_override = nullptr;
    b.W = width; b.H = height;
    b.tiles_x = width / MW_TILE_W; b.tiles_y = height / MW_TILE_H; b.n_tiles = b.tiles_x * b.tiles_y;
    b.env_base = env;
    hipLaunchKernelGGL(geom_kernel_of(e, 64, msaa), dim3(1), dim3(64), 0, st, b, view_flags, msaa, 64, 1);
    uint32_t *keys = nullptr;
    if (e->have_meshes) {
        const size_t need = (size_t)width * height * msaa * 4;
        if (need > e->view_keys_bytes) {
            HIP_TRY(e, hipStreamSynchronize(st));
            if (e->d_view_keys) (void)hipFree(e->d_view_keys);
            e->d_view_keys = nullptr; e->view_keys_bytes = 0;
            HIP_TRY(e, hipMalloc((void **)&e->d_view_keys, need));
            e->view_keys_bytes = need;
        }
        keys = e->d_view_keys;
        HIP_TRY(e, hipMemsetAsync(keys, 0xFF, need, st));
        hipLaunchKernelGGL(mw_view_mesh_kernel, dim3(128), dim3(256), 0, st, width, height, msaa, env, (const float *)b.envhdr, b.mesh_pos, keys);
    }
    hipLaunchKernelGGL(mw_view_raster_kernel, dim3(b.n_tiles), dim3(64), 0, st, env, width, height, msaa, b.max_vis, b.tiles_x,
                       (const float *)b.rec_raster, (const float *)b.rec_shade, (const float *)b.rec_cull, (const int32_t *)b.nvis, (const float *)b.envhdr,
                       b.tex, b.texels, b.mesh_pos, b.mesh_nrm, b.mesh_rgb, b.mesh_uv, (const uint32_t *)keys, d_out, d_depth, e->texel_bytes);
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

int mw_pcg64_draws(uint64_t seed, int32_t n, const int32_t *bounds, double *out)
{
    if (!out || n < 0) return MW_E_INVALID;
    uint64_t s[4];
    pcg64_seed(seed, s);
    mw::Rng r{s[0], s[1], s[2], s[3], 1, 0u, 0u};
    for (int i = 0; i < n; ++i)
        out[i] = (bounds && bounds[i] > 0) ? (double)mw::rng_below(r, (uint32_t)bounds[i]) : mw::rng_double(r);
    return MW_OK;
}

int mw_set_obs_layout(mw_engine *e, int32_t layout)
{
    if (!e) return MW_E_INVALID;
    if (layout != MW_OBS_HWC_U8 && layout != MW_OBS_CWH_U8 && layout != MW_OBS_GREY_F64) return fail(e, MW_E_INVALID, "unknown obs layout %d", layout);
    e->obs_layout = layout;
    return MW_OK;
}

int mw_visible_ents(mw_engine *e, int32_t first_env, int32_t count, uint8_t *d_vis, void *stream)
{
    if (!e || !d_vis) return fail(e, MW_E_INVALID, "null argument");
    ON_DEVICE(e);
    if (first_env < 0 || count <= 0 || first_env + count > e->cfg.num_envs) return fail(e, MW_E_INVALID, "env range out of bounds");
    const size_t lds = (size_t)e->cfg.obs_width * e->cfg.obs_height * e->cfg.msaa * 4;
    if (lds + 1024 > 160 * 1024) return fail(e, MW_E_CAPACITY, "obs frame too large for the in-LDS depth buffer of mw_visible_ents");
    hipStream_t st = (hipStream_t)stream;
    MwArgs b = e->args;
    b.step_override = nullptr;
    b.env_base = first_env;
    // the geometry kernel in proxy mode (view_flags bit 2): room polygons + one tagged proxy box per entity
    {
        const int L = geom_lanes(e), epw = 64 / L;
        hipLaunchKernelGGL(geom_kernel_of(e, L, e->cfg.msaa), dim3((count + epw - 1) / epw), dim3(64), 0, st, b, 4, e->cfg.msaa, L, count);
    }
    if (!e->visible_attr_set) {
        HIP_TRY(e, hipFuncSetAttribute((const void *)mw_visible_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        e->visible_attr_set = true;
    }
    hipLaunchKernelGGL(mw_visible_kernel, dim3(count), dim3(256), lds, st, first_env, e->cfg.obs_width, e->cfg.obs_height,
                       e->cfg.msaa, b.max_vis, e->cfg.max_ents, (const float *)b.rec_raster, (const float *)b.rec_cull, (const int32_t *)b.nvis, d_vis);
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

int mw_check(mw_engine *e, void *stream)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE_SYNC(e);
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    uint32_t st = 0;
    HIP_TRY(e, hipMemcpy(&st, e->args.status, 4, hipMemcpyDeviceToHost));
    if (st & MW_ST_VIS_OVERFLOW) return fail(e, MW_E_OVERFLOW, "more than max_visible=%d visible primitives in some env", e->cfg.max_visible);
    if (st & MW_ST_PLACEMENT_FAIL) return fail(e, MW_E_OVERFLOW, "device-side placement did not converge in some env");
    return MW_OK;
}

int mw_raster_path(const mw_engine *e) { return e ? e->last_raster_path : MW_E_INVALID; }

int mw_debug_set_mesh_frame_seq(mw_engine *e, uint32_t seq)
{
    if (!e) return MW_E_INVALID;
    // (the work lists and the slow-path counters alternate with the sequence number's parity: keep it)
    if ((seq & 1u) != (e->mesh_frame_seq & 1u)) return fail(e, MW_E_INVALID, "mw_debug_set_mesh_frame_seq: the parity of the sequence number must stay");
    e->mesh_frame_seq = seq;
    return MW_OK;
}

int mw_debug_get_slow_heads(mw_engine *e, uint32_t *host_out, void *stream)
{
    if (!e || !host_out) return fail(e, MW_E_INVALID, "null argument");
    if (!e->d_slow_head) return fail(e, MW_E_INVALID, "mw_debug_get_slow_heads: this engine has no mesh path buffers");
    ON_DEVICE(e);
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(e, hipMemcpy(host_out, e->d_slow_head, sizeof(uint32_t) * (size_t)e->cfg.num_envs * e->args.W * e->args.H, hipMemcpyDeviceToHost));
    return MW_OK;
}

int mw_get_list_lengths(mw_engine *e, int32_t first_env, int32_t count, int32_t *host_out, void *stream)
{
    if (!e || !host_out) return fail(e, MW_E_INVALID, "null argument");
    if (first_env < 0 || count <= 0 || first_env + count > e->cfg.num_envs) return fail(e, MW_E_INVALID, "env range out of bounds");
    ON_DEVICE(e);
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(e, hipMemcpy(host_out, e->args.nvis + first_env, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
    return MW_OK;
}

int mw_get_info(mw_engine *e, int32_t *d_health, double *d_ent_pos, int32_t ent_slot, void *stream)
{
    if (!e) return MW_E_INVALID;
    if (!d_health && !d_ent_pos) return fail(e, MW_E_INVALID, "mw_get_info: nothing asked for");
    if (d_ent_pos && (ent_slot < 0 || ent_slot >= e->args.E)) return fail(e, MW_E_INVALID, "mw_get_info: entity slot %d out of range", ent_slot);
    // (the health array exists for the CollectHealth rule only: collecthealth.py:79-100)
    if (d_health && !e->args.health) return fail(e, MW_E_INVALID, "mw_get_info: this engine's task keeps no health (MW_TASK_COLLECT only)");
    ON_DEVICE(e);
    const int N = e->cfg.num_envs;
    hipLaunchKernelGGL(mw_info_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, e->args.E, (const int32_t *)e->args.health,
                       (const double *)e->args.epos, d_ent_pos ? ent_slot : 0, d_health, d_ent_pos);
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

int mw_get_final_info(mw_engine *e, int32_t *d_health, double *d_goal_pos, void *stream)
{
    if (!e) return MW_E_INVALID;
    if (!d_health && !d_goal_pos) return fail(e, MW_E_INVALID, "mw_get_final_info: nothing asked for");
    if (d_health && !e->args.final_health) return fail(e, MW_E_INVALID, "mw_get_final_info: this engine's task keeps no health (MW_TASK_COLLECT only)");
    ON_DEVICE(e);
    const int N = e->cfg.num_envs;
    // (the arrays are component-major like the state: the gather kernel of mw_get_info with slot 0 of a one-slot table)
    hipLaunchKernelGGL(mw_info_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, 1, (const int32_t *)e->args.final_health,
                       (const double *)e->args.final_goal, 0, d_health, d_goal_pos);
    HIP_TRY(e, hipGetLastError());
    return MW_OK;
}

int mw_kernel_time_ms(mw_engine *e, int32_t reset, double *raster_ms, double *setup_ms, int64_t *launches)
{
    if (!e) return MW_E_INVALID;
    ON_DEVICE(e);
    double r = 0, s = 0;
    int64_t n = 0;
    for (auto &ev : e->ev_used) {
        (void)hipEventSynchronize(ev.c);
        float t1 = 0, t2 = 0;
        (void)hipEventElapsedTime(&t1, ev.a, ev.b);
        (void)hipEventElapsedTime(&t2, ev.b, ev.c);
        s += t1; r += t2; ++n;
        e->ev_free.push_back(ev);
    }
    e->ev_used.clear();
    if (raster_ms) *raster_ms = n ? r / n : 0.0;
    if (setup_ms) *setup_ms = n ? s / n : 0.0;
    if (launches) *launches = n;
    e->timing = true;
    e->frame_count = 0;
    e->timing_stride = reset > 0 ? reset : MW_TIMING_STRIDE;
    if (reset < 0) e->timing = false;
    return MW_OK;
}

int mw_abi_version(void) { return MW_ABI_VERSION; }

}  // extern "C"
