// Device-side world generators: MiniWorldEnv.reset (miniworld.py:544-604) for the single
// rectangular-room envs, i.e. the env's _gen_world (hallway.py:55-65, oneroom.py:59-62),
// place_entity's rejection sampling (miniworld.py:872-905), DomainParams.sample of the
// per-episode parameters (miniworld.py:576-585; entity.py:405-407, 505-515).
// Same distributions as the reference, drawn from the engine's Philox stream (the numpy
// PCG64 stream is reproduced host-side only).  Executed by ONE lane.
#pragma once
#include "mw_device.h"
#include "mw_rng.h"

namespace mw {

constexpr double kGenPi = 3.14159265358979323846;

__device__ inline double gen_param(Rng &r, const mw_range &p, bool dr)
{
    return dr ? rng_uniform(r, p.lo, p.hi) : p.def;
}

// circle vs the env's wall segments (math.py:30-62), scalar version
__device__ inline bool gen_hits_wall(const MwArgs &a, int set, double x, double z, double radius)
{
    const double *segs = a.segs + (size_t)set * a.max_segs * 4;
    const int ns = a.nsegs[set];
    for (int i = 0; i < ns; ++i) {
        const double sax = segs[i * 4 + 0], saz = segs[i * 4 + 1], sbx = segs[i * 4 + 2], sbz = segs[i * 4 + 3];
        const double abx = sbx - sax, abz = sbz - saz, apx = x - sax, apz = z - saz;
        double t = (apx * abx + apz * abz) / (abx * abx + abz * abz);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = sax + t * abx - x, dz = saz + t * abz - z;
        if (sqrt(dx * dx + dz * dz) < radius) return true;
    }
    return false;
}

// the same test by the 64 lanes of a wavefront that all hold the same (x, z): segments dealt to the lanes, one ballot
// (the Maze's 256 segments cost one lane ~100 us per placement attempt)
__device__ inline bool gen_hits_wall_wave(const MwArgs &a, int set, double x, double z, double radius, int lane)
{
    const double *segs = a.segs + (size_t)set * a.max_segs * 4;
    const int ns = a.nsegs[set];
    bool hit = false;
    for (int i = lane; i < ns; i += 64) {
        const double sax = segs[i * 4 + 0], saz = segs[i * 4 + 1], sbx = segs[i * 4 + 2], sbz = segs[i * 4 + 3];
        const double abx = sbx - sax, abz = sbz - saz, apx = x - sax, apz = z - saz;
        double t = (apx * abx + apz * abz) / (abx * abx + abz * abz);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = sax + t * abx - x, dz = saz + t * abz - z;
        hit |= sqrt(dx * dx + dz * dz) < radius;
    }
    return __any(hit) != 0;
}

// place_entity in the rectangular room gen_args[0..3] (miniworld.py:872-905); lx/hx narrow
// the sampled x range like the min_x / max_x keyword arguments do.
__device__ inline bool gen_place(const MwArgs &a, int env, int set, Rng &r, double radius, int n_placed,
                                 double lx, double hx, double &ox, double &oz)
{
    const double rx0 = a.gen_args[0], rx1 = a.gen_args[1], rz0 = a.gen_args[2], rz1 = a.gen_args[3];
    for (int attempt = 0; attempt < 4096; ++attempt) {
        // reference stream: np_random.choice(len(rooms), p=room_probs) draws one double even for a single
        // room, and uniform(low=[x, 0, z], high=[x, 0, z]) one for the y component (miniworld.py:873-895)
        if (rng_is_pcg(r)) (void)rng_double(r);
        const double x = rng_uniform(r, lx - radius, hx + radius);
        if (rng_is_pcg(r)) (void)rng_uniform(r, 0.0, 0.0);
        const double z = rng_uniform(r, rz0 - radius, rz1 + radius);
        // Room.point_inside (miniworld.py:272-284): strictly inside every edge
        if (!(x > rx0 && x < rx1 && z > rz0 && z < rz1)) continue;
        if (gen_hits_wall(a, set, x, z, radius)) continue;
        bool hit = false;
        for (int s = 0; s < n_placed; ++s) {
            const double dx = a.epos[((size_t)0 * a.E + s) * a.N + env] - x;
            const double dz = a.epos[((size_t)2 * a.E + s) * a.N + env] - z;
            hit |= sqrt(dx * dx + dz * dz) < radius + a.egeom[((size_t)7 * a.E + s) * a.N + env];
        }
        if (hit) continue;
        ox = x; oz = z;
        return true;
    }
    atomicOr(a.status, MW_ST_PLACEMENT_FAIL);
    ox = 0.5 * (rx0 + rx1); oz = 0.5 * (rz0 + rz1);
    return false;
}

__device__ inline void gen_store_box(const MwArgs &a, int env, int slot, double x, double z, double dir, double size,
                                     const double col[3])
{
    const size_t N = a.N, E = a.E;
    a.ekind[(size_t)slot * N + env] = MW_ENT_BOX;
    a.emesh[(size_t)slot * N + env] = -1;
    a.estatic[(size_t)slot * N + env] = 0;
    a.epos[((size_t)0 * E + slot) * N + env] = x;
    a.epos[((size_t)1 * E + slot) * N + env] = 0.0;
    a.epos[((size_t)2 * E + slot) * N + env] = z;
    a.edir[(size_t)slot * N + env] = dir;
    for (int k = 0; k < 3; ++k) a.egeom[((size_t)k * E + slot) * N + env] = size;
    for (int k = 0; k < 3; ++k) a.egeom[((size_t)(3 + k) * E + slot) * N + env] = col[k];
    a.egeom[((size_t)6 * E + slot) * N + env] = 1.0;
    a.egeom[((size_t)7 * E + slot) * N + env] = sqrt(size * size + size * size) / 2.0;   // Box.radius entity.py:401
    a.egeom[((size_t)8 * E + slot) * N + env] = size;
}

// ---------------------------------------------------------------- Maze (maze.py:73-153)
// Per-env geometry is written straight into the env's polygon / segment set in the layout and
// order Room._gen_static_data + Room._render produce on the host (miniworld.py:286-434): cells
// row-major, then the connecting rooms in carving order; per room floor, ceiling, kept walls.
// gen_tab: [0] rows, [1] cols, [2] room_size, [3] gap, [4] wall_height, [5..7] texture ids of
// floor / ceiling / wall; gen_colors[0..5]: TEX_DENSITY / texture size (u, v) for the same three.

struct MazeRect { double x0, x1, z0, z1; };     // min_x, max_x, min_z, max_z

struct RoomTex {            // textures of one emitted room: ids and TEX_DENSITY / size per axis
    int floor, ceil, wall;
    double fu, fv, cu, cv, wu, wv;
    double height;
    bool ceiling;
};

__device__ inline void emit_room(const MwArgs &a, int set, const RoomTex &rt, const double px[4], const double pz[4],
                                 unsigned keep_walls, int &np, int &ns)
{
    mw_poly *polys = const_cast<mw_poly *>(a.polys) + (size_t)set * a.max_polys;
    double *segs = const_cast<double *>(a.segs) + (size_t)set * a.max_segs * 4;
    const double h = rt.height;
    const int tex_f = rt.floor, tex_c = rt.ceil, tex_w = rt.wall;
    {   // floor: the outline itself, normal +Y (miniworld.py:408-415), texcoords = (x, z) * density
        mw_poly &q = polys[np++];
        for (int k = 0; k < 4; ++k) {
            q.v[k][0] = (float)px[k]; q.v[k][1] = 0.0f; q.v[k][2] = (float)pz[k];
            q.uv[k][0] = (float)(px[k] * rt.fu); q.uv[k][1] = (float)(pz[k] * rt.fv);
        }
        q.n[0] = 0.0f; q.n[1] = 1.0f; q.n[2] = 0.0f; q.nv = 4; q.tex = tex_f; q.rgb[0] = q.rgb[1] = q.rgb[2] = 1.0f;
    }
    if (rt.ceiling) {   // ceiling: flipped outline at wall height, normal -Y (:304-306, 418-425)
        mw_poly &q = polys[np++];
        for (int k = 0; k < 4; ++k) {
            const double x = px[3 - k], z = pz[3 - k];
            q.v[k][0] = (float)x; q.v[k][1] = (float)(0.0 + h * 1.0); q.v[k][2] = (float)z;
            q.uv[k][0] = (float)(x * rt.cu); q.uv[k][1] = (float)(z * rt.cv);
        }
        q.n[0] = 0.0f; q.n[1] = -1.0f; q.n[2] = 0.0f; q.nv = 4; q.tex = tex_c; q.rgb[0] = q.rgb[1] = q.rgb[2] = 1.0f;
    }
    for (int w = 0; w < 4; ++w) {
        if (!((keep_walls >> w) & 1u)) continue;
        const double ax = px[w], az = pz[w], bx0 = px[(w + 1) & 3], bz0 = pz[(w + 1) & 3];
        const double dx = bx0 - ax, dz = bz0 - az;
        const double width = sqrt(dx * dx + dz * dz);
        const double sx = dx / width, sz = dz / width;
        const double bx = ax + width * sx, bz = az + width * sz;
        mw_poly &q = polys[np++];
        q.v[0][0] = (float)ax; q.v[0][1] = 0.0f;     q.v[0][2] = (float)az;
        q.v[1][0] = (float)ax; q.v[1][1] = (float)h; q.v[1][2] = (float)az;
        q.v[2][0] = (float)bx; q.v[2][1] = (float)h; q.v[2][2] = (float)bz;
        q.v[3][0] = (float)bx; q.v[3][1] = 0.0f;     q.v[3][2] = (float)bz;
        const float u1 = (float)((0 + width) * rt.wu), v1 = (float)((0 + h) * rt.wv);
        q.uv[0][0] = 0.0f; q.uv[0][1] = 0.0f; q.uv[1][0] = 0.0f; q.uv[1][1] = v1;
        q.uv[2][0] = u1;   q.uv[2][1] = v1;   q.uv[3][0] = u1;   q.uv[3][1] = 0.0f;
        // normal = -cross(b - a, Y) / |.|   (miniworld.py:335-336)
        const double ex = bx - ax, ez = bz - az, len = sqrt(ez * ez + ex * ex);
        q.n[0] = (float)(-(-ez) / len); q.n[1] = 0.0f; q.n[2] = (float)(-(ex) / len);
        q.nv = 4 | MW_POLY_QUAD; q.tex = tex_w; q.rgb[0] = q.rgb[1] = q.rgb[2] = 1.0f;
        segs[ns * 4 + 0] = bx; segs[ns * 4 + 1] = bz; segs[ns * 4 + 2] = ax; segs[ns * 4 + 3] = az;   // [s_p1, s_p0]
        ++ns;
    }
}

__device__ inline MazeRect maze_cell(const MwArgs &a, int i, int j)
{
    const double pitch = a.gt->gen_tab[2] + a.gt->gen_tab[3];
    MazeRect r;
    r.x0 = i * pitch; r.x1 = r.x0 + a.gt->gen_tab[2];
    r.z0 = j * pitch; r.z1 = r.z0 + a.gt->gen_tab[2];
    return r;
}

// outline of the room connecting cell A to its neighbour B in direction d (0 east, 1 west,
// 2 north, 3 south) — connect_rooms' [c, b, a, d] (miniworld.py:807-821) for facing rect rooms
__device__ inline void maze_link_outline(const MazeRect &A, const MazeRect &B, int d, double px[4], double pz[4])
{
    switch (d) {
    case 0: px[0] = B.x0; pz[0] = A.z0; px[1] = A.x1; pz[1] = A.z0; px[2] = A.x1; pz[2] = A.z1; px[3] = B.x0; pz[3] = A.z1; break;
    case 1: px[0] = B.x1; pz[0] = A.z1; px[1] = A.x0; pz[1] = A.z1; px[2] = A.x0; pz[2] = A.z0; px[3] = B.x1; pz[3] = A.z0; break;
    case 2: px[0] = A.x0; pz[0] = B.z1; px[1] = A.x0; pz[1] = A.z0; px[2] = A.x1; pz[2] = A.z0; px[3] = A.x1; pz[3] = B.z1; break;
    default: px[0] = A.x1; pz[0] = B.z0; px[1] = A.x1; pz[1] = A.z1; px[2] = A.x0; pz[2] = A.z1; px[3] = A.x0; pz[3] = B.z0; break;
    }
}

#define MW_GEN_WS_BYTES 400     // per-env scratch of the generators (LDS, provided by the caller)

// Called by all 64 lanes of one wavefront: lane 0 carves and places (the random stream is sequential), the
// 127 rooms are emitted one per lane.
__device__ inline void gen_maze(const MwArgs &a, int env, int set, Rng &r, unsigned char *ws, int lane, double &bx, double &bz,
                                double &bdir, double &ax, double &az, double &adir)
{
    const int rows = (int)a.gt->gen_tab[0], cols = (int)a.gt->gen_tab[1];
    const int ncell = rows * cols;
    // direction d: (dj, di) of maze.py:113  (0,1) (0,-1) (-1,0) (1,0); wall of A opened / of B opened
    const int DI[4] = {1, -1, 0, 0}, DJ[4] = {0, 0, -1, 1};
    const int OPEN_A[4] = {0, 2, 1, 3}, OPEN_B[4] = {2, 0, 3, 1};
    // workspace in LDS (keeps the kernels' register budgets small): 6 arrays of 64 bytes
    unsigned char *open = ws;               // per cell: bit w = wall w replaced by a portal
    unsigned char *link_cell = ws + 64, *link_dir = ws + 128;
    unsigned char *st_cell = ws + 192, *st_perm = ws + 256, *st_next = ws + 320;
    int *s_nlink = reinterpret_cast<int *>(ws + 384);
    if (lane == 0) {
    unsigned long long visited = 0ull;
    for (int k = 0; k < ncell; ++k) open[k] = 0;
    int nlink = 0, sp = 0;
    // recursive backtracker (maze.py:100-149), explicit stack; the visiting order of the 4
    // neighbours is drawn without replacement when a cell is first entered
    auto enter = [&](int cell) {
        visited |= 1ull << cell;
        int pool[4] = {0, 1, 2, 3};
        unsigned perm = 0;
        for (int k = 4; k >= 1; --k) {
            const int idx = (int)rng_below(r, (uint32_t)k);
            perm |= (unsigned)pool[idx] << (2 * (4 - k));
            for (int m = idx; m < k - 1; ++m) pool[m] = pool[m + 1];
        }
        st_cell[sp] = (unsigned char)cell; st_perm[sp] = (unsigned char)perm; st_next[sp] = 0;
        ++sp;
    };
    enter(0);
    while (sp > 0) {
        const int top = sp - 1;
        if (st_next[top] >= 4) { --sp; continue; }
        const int d = (st_perm[top] >> (2 * st_next[top])) & 3;
        st_next[top]++;
        const int cell = st_cell[top], i = cell % cols, j = cell / cols;
        const int ni = i + DI[d], nj = j + DJ[d];
        if (ni < 0 || ni >= cols || nj < 0 || nj >= rows) continue;
        const int ncellidx = nj * cols + ni;
        if ((visited >> ncellidx) & 1ull) continue;
        open[cell] |= (unsigned char)(1u << OPEN_A[d]);
        open[ncellidx] |= (unsigned char)(1u << OPEN_B[d]);
        link_cell[nlink] = (unsigned char)cell; link_dir[nlink] = (unsigned char)d; ++nlink;
        enter(ncellidx);
    }
    *s_nlink = nlink;
    }       // lane 0
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int nlink = *s_nlink;
    // ---- texture variants: with domain randomisation every room draws its wall, floor and ceiling variant, in that order,
    // rooms in list order (Room._gen_static_data with an rng, miniworld.py:295-297, run for all rooms inside the first
    // place_entity: after the carving, before the placement; a texture with one variant draws nothing, opengl.py:134-138).
    // Lane 0 draws; the picks (a nibble each) go where the carving's stack was.
    const bool tex_var = a.domain_rand != 0 && (a.gt->tex_nvar[0] > 1 || a.gt->tex_nvar[1] > 1 || a.gt->tex_nvar[2] > 1);
    unsigned char *picks = ws + 192;        // 3 nibbles per room, 127 rooms: 191 bytes
    if (tex_var) {
        if (lane == 0) {
            for (int i = 0; i < 192; ++i) picks[i] = 0;
            for (int room = 0; room < ncell + nlink; ++room)
                for (int k = 0; k < 3; ++k) {
                    const uint32_t nv = (uint32_t)a.gt->tex_nvar[k];
                    const uint32_t pk = nv > 1u ? rng_below(r, nv) : 0u;
                    picks[(3 * room + k) >> 1] |= (unsigned char)(pk << (4 * ((3 * room + k) & 1)));
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- geometry: one room per lane, offsets from an ordered prefix sum over the rooms ------
    RoomTex rt;
    rt.floor = (int)a.gt->gen_tab[5]; rt.ceil = (int)a.gt->gen_tab[6]; rt.wall = (int)a.gt->gen_tab[7];
    rt.fu = a.gt->gen_colors[0]; rt.fv = a.gt->gen_colors[1]; rt.cu = a.gt->gen_colors[2]; rt.cv = a.gt->gen_colors[3];
    rt.wu = a.gt->gen_colors[4]; rt.wv = a.gt->gen_colors[5];
    rt.height = a.gt->gen_tab[4]; rt.ceiling = true;
    {
        const int nroom = ncell + nlink;
        int np_base = 0, ns_base = 0;
        for (int r0 = 0; r0 < nroom; r0 += 64) {
            const int room = r0 + lane;
            double px[4] = {0, 0, 0, 0}, pz[4] = {0, 0, 0, 0};
            unsigned keep = 0u;
            if (room < ncell) {
                const MazeRect c = maze_cell(a, room % cols, room / cols);
                px[0] = c.x1; px[1] = c.x1; px[2] = c.x0; px[3] = c.x0;       // add_rect_room outline
                pz[0] = c.z1; pz[1] = c.z0; pz[2] = c.z0; pz[3] = c.z1;
                keep = (~(unsigned)open[room]) & 15u;
            } else if (room < nroom) {
                const int k = room - ncell, cell = link_cell[k], d = link_dir[k];
                const MazeRect A = maze_cell(a, cell % cols, cell / cols);
                const MazeRect B = maze_cell(a, cell % cols + DI[d], cell / cols + DJ[d]);
                maze_link_outline(A, B, d, px, pz);
                keep = 5u;          // walls 1 and 3 are portals (miniworld.py:836-837)
            }
            const int nw = __popc(keep);
            int cnt = room < nroom ? (2 + nw) | (nw << 16) : 0;       // polygons | segments << 16
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            int np = np_base + ((incl - cnt) & 0xFFFF), ns = ns_base + ((incl - cnt) >> 16);
            if (room < nroom) {
                RoomTex rr = rt;
                if (tex_var) {
                    int pk[3];
                    for (int k = 0; k < 3; ++k) pk[k] = (picks[(3 * room + k) >> 1] >> (4 * ((3 * room + k) & 1))) & 15;
                    rr.wall = a.gt->tex_var_id[0][pk[0]]; rr.wu = a.gt->tex_var_scale[0][pk[0]][0]; rr.wv = a.gt->tex_var_scale[0][pk[0]][1];
                    rr.floor = a.gt->tex_var_id[1][pk[1]]; rr.fu = a.gt->tex_var_scale[1][pk[1]][0]; rr.fv = a.gt->tex_var_scale[1][pk[1]][1];
                    rr.ceil = a.gt->tex_var_id[2][pk[2]]; rr.cu = a.gt->tex_var_scale[2][pk[2]][0]; rr.cv = a.gt->tex_var_scale[2][pk[2]][1];
                }
                emit_room(a, set, rr, px, pz, keep, np, ns);
            }
            const int tot = __shfl(incl, 63);
            np_base += tot & 0xFFFF; ns_base += tot >> 16;
        }
        if (lane == 0) {
            const_cast<int32_t *>(a.npolys)[set] = np_base;
            if (a.occ_valid) a.occ_valid[set] = 0;
            const_cast<int32_t *>(a.nsegs)[set] = ns_base;
        }
    }
    __threadfence();                // the placement is tested against the segments the lanes just wrote
    __builtin_amdgcn_wave_barrier();
    // ---- placement: box then agent, room drawn with probability ~ area (miniworld.py:872-905) --
    // All 64 lanes run it: the random stream is a pure function of its state, so after lane 0 (which alone advanced
    // it while carving) has handed its state over, every lane draws the same candidates and takes the same decisions,
    // and the one expensive step — the candidate against the env's 256 wall segments — is shared out (ballot).
    {
        const uint32_t a_lo = (uint32_t)__shfl((int)(uint32_t)r.a, 0), a_hi = (uint32_t)__shfl((int)(uint32_t)(r.a >> 32), 0);
        const uint32_t b_lo = (uint32_t)__shfl((int)(uint32_t)r.b, 0), b_hi = (uint32_t)__shfl((int)(uint32_t)(r.b >> 32), 0);
        r.a = ((uint64_t)a_hi << 32) | a_lo; r.b = ((uint64_t)b_hi << 32) | b_lo;
        r.has32 = (uint32_t)__shfl((int)r.has32, 0); r.buf32 = (uint32_t)__shfl((int)r.buf32, 0);
    }
    const double cell_area = a.gt->gen_tab[2] * a.gt->gen_tab[2], link_area = a.gt->gen_tab[2] * a.gt->gen_tab[3];
    const double total = ncell * cell_area + nlink * link_area;
    const double radii[2] = {sqrt(0.8 * 0.8 + 0.8 * 0.8) / 2.0, a.agent_radius};
    double out[2][2];
    for (int who = 0; who < 2; ++who) {
        const double rad = radii[who];
        bool placed = false;
        for (int attempt = 0; attempt < 100000 && !placed; ++attempt) {
            const double u = rng_double(r) * total;
            MazeRect rr;
            if (u < ncell * cell_area) {
                int cell = (int)(u / cell_area);
                cell = cell < ncell ? cell : ncell - 1;
                rr = maze_cell(a, cell % cols, cell / cols);
            } else {
                int k = (int)((u - ncell * cell_area) / link_area);
                k = k < nlink ? k : nlink - 1;
                const int cell = link_cell[k], d = link_dir[k];
                const MazeRect A = maze_cell(a, cell % cols, cell / cols);
                const MazeRect B = maze_cell(a, cell % cols + DI[d], cell / cols + DJ[d]);
                double px[4], pz[4];
                maze_link_outline(A, B, d, px, pz);
                rr.x0 = fmin(fmin(px[0], px[1]), fmin(px[2], px[3])); rr.x1 = fmax(fmax(px[0], px[1]), fmax(px[2], px[3]));
                rr.z0 = fmin(fmin(pz[0], pz[1]), fmin(pz[2], pz[3])); rr.z1 = fmax(fmax(pz[0], pz[1]), fmax(pz[2], pz[3]));
            }
            const double x = rng_uniform(r, rr.x0 - rad, rr.x1 + rad);
            if (rng_is_pcg(r)) (void)rng_uniform(r, 0.0, 0.0);       // the y component of the 3-vector draw
            const double z = rng_uniform(r, rr.z0 - rad, rr.z1 + rad);
            if (!(x > rr.x0 && x < rr.x1 && z > rr.z0 && z < rr.z1)) continue;        // Room.point_inside
            if (gen_hits_wall_wave(a, set, x, z, rad, lane)) continue;
            if (who == 1) {
                const double dx = out[0][0] - x, dz = out[0][1] - z;
                if (sqrt(dx * dx + dz * dz) < rad + radii[0]) continue;
            }
            out[who][0] = x; out[who][1] = z;
            placed = true;
        }
        if (!placed) { atomicOr(a.status, MW_ST_PLACEMENT_FAIL); out[who][0] = 1.5; out[who][1] = 1.5; }
        if (who == 0) bdir = rng_uniform(r, -kGenPi, kGenPi); else adir = rng_uniform(r, -kGenPi, kGenPi);
    }
    bx = out[0][0]; bz = out[0][1]; ax = out[1][0]; az = out[1][1];
}

// ---------------------------------------------------------------- placement programs (MW_GEN_PROGRAM)
// include/mwengine.h: mw_gen_program.  One lane runs the env's program on its random stream.

// Room.point_inside (miniworld.py:272-284): np.sum(edge_norms * (p - outline), axis=1) > 0 for every edge; the y terms
// are +-0 * 0, so a row's sum is nx * dx + nz * dz (two rounded products, one rounded sum: no fma)
__device__ inline bool prog_point_inside(const mw_prog_room &rm, double x, double z)
{
    bool in = true;
    for (int k = 0; k < rm.nverts; ++k) {
        const double t = rm.nx[k] * (x - rm.ox[k]) + rm.nz[k] * (z - rm.oz[k]);
        in &= t > 0.0;
    }
    return in;
}

// MiniWorldEnv.intersect as place_entity uses it (miniworld.py:937-963): walls, then every entity already in the list
__device__ inline bool prog_blocked(const MwArgs &a, int env, int set, unsigned long long placed, bool agent_placed,
                                    double agent_x, double agent_z, double x, double z, double radius)
{
    if (gen_hits_wall(a, set, x, z, radius)) return true;
    for (int s = 0; s < a.E; ++s) {
        if (!((placed >> s) & 1ull)) continue;
        const double dx = a.epos[((size_t)0 * a.E + s) * a.N + env] - x;
        const double dz = a.epos[((size_t)2 * a.E + s) * a.N + env] - z;
        if (sqrt(dx * dx + dz * dz) < radius + a.egeom[((size_t)7 * a.E + s) * a.N + env]) return true;
    }
    if (agent_placed) {
        const double dx = agent_x - x, dz = agent_z - z;
        if (sqrt(dx * dx + dz * dz) < radius + a.agent_radius) return true;
    }
    return false;
}

// Room._gen_static_data of every room with an rng (miniworld.py:295-297, opengl.py:134-138): per room the wall, floor
// and ceiling variants are drawn in that order; the template polygons are then re-emitted into the env's own set
// with the variants' ids and texture coordinates = (float)(metres * TEX_DENSITY / size) (miniworld.py:82-119)
__device__ inline void prog_static_data(const MwArgs &a, int env, int set, Rng &r)
{
    const MwProgram &P = *a.prog;
    if (!a.domain_rand || a.shared_geom) return;
    unsigned char pick[MW_PROG_MAX_ROOMS][3];
    for (int i = 0; i < P.p.n_rooms; ++i) {
        const mw_prog_room &rm = P.p.rooms[i];
        const int t[3] = {rm.wall_tex, rm.floor_tex, rm.ceil_tex};
        for (int k = 0; k < 3; ++k) pick[i][k] = (unsigned char)(P.p.tex_nvar[t[k]] > 1 ? rng_below(r, (uint32_t)P.p.tex_nvar[t[k]]) : 0u);
    }
    mw_poly *dst = const_cast<mw_poly *>(a.polys) + (size_t)set * a.max_polys;
    for (int p = 0; p < P.n_polys; ++p) {
        mw_poly q = P.polys[p];
        const int room = P.poly_room[p];
        if (room >= 0) {
            const mw_prog_room &rm = P.p.rooms[room];
            const int surf = P.poly_surf[p];
            const int t = surf == 0 ? rm.wall_tex : (surf == 1 ? rm.floor_tex : rm.ceil_tex);
            const int v = pick[room][surf];
            q.tex = P.p.tex_var_id[t][v];
            const double ku = P.p.tex_var_scale[t][v][0], kv = P.p.tex_var_scale[t][v][1];
            const int nv = q.nv & 0xFF;
            for (int k = 0; k < nv; ++k) {
                q.uv[k][0] = (float)(P.poly_m[((size_t)p * 4 + k) * 2 + 0] * ku);
                q.uv[k][1] = (float)(P.poly_m[((size_t)p * 4 + k) * 2 + 1] * kv);
            }
        }
        dst[p] = q;
    }
    double *sd = const_cast<double *>(a.segs) + (size_t)set * a.max_segs * 4;
    for (int i = 0; i < P.n_segs * 4; ++i) sd[i] = P.segs[i];
    const_cast<int32_t *>(a.npolys)[set] = P.n_polys;
    if (a.occ_valid) a.occ_valid[set] = 0;
    const_cast<int32_t *>(a.nsegs)[set] = P.n_segs;
    __threadfence();        // the placements below test against these segments
}

// place_entity's rejection sampling (miniworld.py:872-905) for one PLACE op: room (given or drawn by area), a 3-vector
// draw whose y is thrown away, Room.point_inside, MiniWorldEnv.intersect against walls and the entities in the list
__device__ inline void prog_sample_position(const MwArgs &a, int env, int set, Rng &r, const mw_prog_op &op, double rad,
                                            unsigned long long placed, bool agent_placed, double agent_x, double agent_z,
                                            double &x, double &z)
{
    const mw_gen_program &g = a.prog->p;
    for (int attempt = 0; attempt < 100000; ++attempt) {
        int ri = op.room;
        if (ri < 0) {       // np_random.choice(len(rooms), p=room_probs): one double, searchsorted right
            ri = 0;
            if (rng_is_pcg(r) || g.n_rooms > 1) {
                const double u = rng_double(r);
                while (ri < g.n_rooms - 1 && !(u < g.rooms[ri].cdf)) ++ri;
            }
        }
        const mw_prog_room &rm = g.rooms[ri];
        const double lx = (op.flags & 1) ? op.lx : rm.min_x, hx = (op.flags & 2) ? op.hx : rm.max_x;
        const double lz = (op.flags & 4) ? op.lz : rm.min_z, hz = (op.flags & 8) ? op.hz : rm.max_z;
        x = rng_uniform(r, lx - rad, hx + rad);
        if (rng_is_pcg(r)) (void)rng_uniform(r, 0.0, 0.0);        // the y component of the 3-vector draw
        z = rng_uniform(r, lz - rad, hz + rad);
        if (!prog_point_inside(rm, x, z)) continue;
        if (prog_blocked(a, env, set, placed, agent_placed, agent_x, agent_z, x, z, rad)) continue;
        return;
    }
    atomicOr(a.status, MW_ST_PLACEMENT_FAIL);
}

__device__ inline void gen_program(const MwArgs &a, int env, int set, Rng &r, double &ax, double &az, double &adir)
{
    const MwProgram &P = *a.prog;
    const mw_gen_program &g = P.p;
    const size_t N = a.N, E = a.E;
    // the entity table as the constructors leave it
    for (int s = 0; s < (int)E; ++s) {
        const bool have = s < g.n_ents;
        a.ekind[(size_t)s * N + env] = have ? g.ent_kind[s] : MW_ENT_NONE;
        a.emesh[(size_t)s * N + env] = have ? g.ent_mesh[s] : -1;
        a.estatic[(size_t)s * N + env] = have ? g.ent_static[s] : 0;
        a.edir[(size_t)s * N + env] = have ? g.ent_dir[s] : 0.0;
        for (int k = 0; k < 3; ++k) a.epos[((size_t)k * E + s) * N + env] = have ? g.ent_pos[s][k] : 0.0;
        for (int k = 0; k < 9; ++k) a.egeom[((size_t)k * E + s) * N + env] = have ? g.ent_geom[s][k] : 0.0;
    }
    unsigned long long placed = 0ull;
    bool agent_placed = false, static_done = false;
    int coin = -1;
    double dir_reg = 0.0;
    for (int i = 0; i < g.n_ops; ++i) {
        const mw_prog_op &op = g.ops[i];
        if (op.cond >= 0 && op.cond != coin) continue;
        switch (op.op) {
        case MW_OP_COIN: coin = (int)rng_below(r, (uint32_t)op.slot); break;
        case MW_OP_DRAW_DIR: dir_reg = rng_uniform(r, -op.dir, op.dir); break;
        case MW_OP_BOX_SIZE: {
            const double size = rng_uniform(r, op.a, op.b);
            for (int k = 0; k < 3; ++k) a.egeom[((size_t)k * E + op.slot) * N + env] = size;
            a.egeom[((size_t)7 * E + op.slot) * N + env] = sqrt(size * size + size * size) / 2.0;      // Box.radius entity.py:401
            a.egeom[((size_t)8 * E + op.slot) * N + env] = size;
            break;
        }
        case MW_OP_COLOR: {
            const int c = (int)rng_below(r, 6u);
            if (op.room == 0) for (int k = 0; k < 3; ++k) a.egeom[((size_t)(3 + k) * E + op.slot) * N + env] = g.colors[c][k];
            else a.emesh[(size_t)op.slot * N + env] = op.flags + c;
            break;
        }
        case MW_OP_APPEND: placed |= 1ull << op.slot; break;
        case MW_OP_FIXED:
        case MW_OP_PLACE: {
            if (!static_done) { prog_static_data(a, env, set, r); static_done = true; }
            const bool agent = op.slot < 0;
            const double rad = agent ? a.agent_radius : a.egeom[((size_t)7 * E + op.slot) * N + env];
            double x = op.lx, z = op.lz, y = 0.0;
            if (op.op == MW_OP_FIXED) {
                y = op.a;
            } else {
                prog_sample_position(a, env, set, r, op, rad, placed, agent_placed, ax, az, x, z);
            }
            const double dir = op.dir_mode == 1 ? op.dir : (op.dir_mode == 2 ? dir_reg : rng_uniform(r, -kGenPi, kGenPi));
            if (agent) {
                ax = x; az = z; adir = dir; agent_placed = true;
            } else {
                a.epos[((size_t)0 * E + op.slot) * N + env] = x;
                a.epos[((size_t)1 * E + op.slot) * N + env] = y;
                a.epos[((size_t)2 * E + op.slot) * N + env] = z;
                a.edir[(size_t)op.slot * N + env] = dir;
                placed |= 1ull << op.slot;
            }
            break;
        }
        default: break;
        }
    }
    if (!static_done) prog_static_data(a, env, set, r);
}

// CollectHealth (collecthealth.py:86-90): the consumed kit leaves the entity list and is placed again at its END —
// slots are list positions, so the slots behind it move down by one and the new kit takes the last one — with
// place_entity's draws from the env's stream.  One lane, after the frame's primitives were set up.
__device__ inline void collect_respawn(const MwArgs &a, int env, int set, int k, double agent_x, double agent_z)
{
    const size_t N = a.N, E = a.E;
    const int n = a.prog->p.n_ents, last = n - 1;
    int32_t kind = a.ekind[(size_t)k * N + env], mesh = a.emesh[(size_t)k * N + env], stat = a.estatic[(size_t)k * N + env];
    double geom[9];
    for (int j = 0; j < 9; ++j) geom[j] = a.egeom[((size_t)j * E + k) * N + env];
    for (int s = k; s < last; ++s) {
        a.ekind[(size_t)s * N + env] = a.ekind[(size_t)(s + 1) * N + env];
        a.emesh[(size_t)s * N + env] = a.emesh[(size_t)(s + 1) * N + env];
        a.estatic[(size_t)s * N + env] = a.estatic[(size_t)(s + 1) * N + env];
        a.edir[(size_t)s * N + env] = a.edir[(size_t)(s + 1) * N + env];
        for (int j = 0; j < 3; ++j) a.epos[((size_t)j * E + s) * N + env] = a.epos[((size_t)j * E + s + 1) * N + env];
        for (int j = 0; j < 9; ++j) a.egeom[((size_t)j * E + s) * N + env] = a.egeom[((size_t)j * E + s + 1) * N + env];
    }
    a.ekind[(size_t)last * N + env] = kind; a.emesh[(size_t)last * N + env] = mesh; a.estatic[(size_t)last * N + env] = stat;
    for (int j = 0; j < 9; ++j) a.egeom[((size_t)j * E + last) * N + env] = geom[j];
    __threadfence();
    Rng r = rng_load(a.rng, a.N, env);
    mw_prog_op op{};
    op.op = MW_OP_PLACE; op.slot = last; op.room = -1; op.cond = -1; op.flags = 0;
    unsigned long long placed = 0ull;
    for (int s = 0; s < last; ++s)
        if (a.ekind[(size_t)s * N + env] != MW_ENT_NONE) placed |= 1ull << s;
    double x = 0.0, z = 0.0;
    prog_sample_position(a, env, set, r, op, geom[7], placed, true, agent_x, agent_z, x, z);
    a.epos[((size_t)0 * E + last) * N + env] = x;
    a.epos[((size_t)1 * E + last) * N + env] = 0.0;
    a.epos[((size_t)2 * E + last) * N + env] = z;
    a.edir[(size_t)last * N + env] = rng_uniform(r, -kGenPi, kGenPi);
    rng_store(a.rng, a.N, env, r);
}

// One full reset of env `env`.  Writes every per-env state array.
// Called by the 64 lanes of one wavefront; everything but the Maze's room emission runs on lane 0.
__device__ inline void generate_world(const MwArgs &a, int env, unsigned char *ws, int lane)
{
    if (a.generator != MW_GEN_MAZE && lane != 0) return;
    const int set = a.shared_geom ? 0 : env;
    const bool dr = a.domain_rand != 0;
    Rng r = rng_load(a.rng, a.N, env);
    const size_t N = a.N;
    for (int s = 0; s < a.E; ++s) a.ekind[(size_t)s * N + env] = MW_ENT_NONE;
    double ax = 0, az = 0, adir = 0;
    // Room._gen_static_data with an rng (miniworld.py:295-297): wall, floor, ceiling variant drawn in that
    // order (opengl.py:136-138); the room is re-emitted into this env's own set.  The reference runs it
    // inside the first place_entity (miniworld.py:856-857), i.e. after whatever _gen_world drew before.
    auto pick_textures = [&]() {
        if (!(a.gt->tex_nvar[0] > 0 && !a.shared_geom && a.generator != MW_GEN_MAZE)) return;
        int pick[3];
        for (int k = 0; k < 3; ++k) pick[k] = a.gt->tex_nvar[k] > 1 ? (int)rng_below(r, (uint32_t)a.gt->tex_nvar[k]) : 0;
        RoomTex rt;
        rt.wall = a.gt->tex_var_id[0][pick[0]]; rt.wu = a.gt->tex_var_scale[0][pick[0]][0]; rt.wv = a.gt->tex_var_scale[0][pick[0]][1];
        rt.floor = a.gt->tex_var_id[1][pick[1]]; rt.fu = a.gt->tex_var_scale[1][pick[1]][0]; rt.fv = a.gt->tex_var_scale[1][pick[1]][1];
        rt.ceil = a.gt->tex_var_id[2][pick[2]]; rt.cu = a.gt->tex_var_scale[2][pick[2]][0]; rt.cv = a.gt->tex_var_scale[2][pick[2]][1];
        rt.height = a.gt->room_wall_height; rt.ceiling = !a.gt->room_no_ceiling;
        const double px[4] = {a.gen_args[1], a.gen_args[1], a.gen_args[0], a.gen_args[0]};
        const double pz[4] = {a.gen_args[3], a.gen_args[2], a.gen_args[2], a.gen_args[3]};
        int np = 0, ns = 0;
        emit_room(a, set, rt, px, pz, 15u, np, ns);
        const_cast<int32_t *>(a.npolys)[set] = np;
        if (a.occ_valid) a.occ_valid[set] = 0;
        const_cast<int32_t *>(a.nsegs)[set] = ns;
    };
    // PickupObjects draws its first object's kind and colour before the first placement; the Philox
    // stream keeps its historical order (textures first)
    const bool tex_late = rng_is_pcg(r) && a.generator == MW_GEN_PICKUP;
    if (!tex_late) pick_textures();
    if (a.generator == MW_GEN_HALLWAY || a.generator == MW_GEN_ONEROOM || a.generator == MW_GEN_MAZE) {
        // the red box (hallway.py:59, oneroom.py:61, maze.py:151), then the agent
        const double size = a.generator == MW_GEN_MAZE ? 0.8 : a.gen_args[7];
        double col[3] = {1.0, 0.0, 0.0};                       // COLORS["red"] entity.py:31
        if (a.generator == MW_GEN_MAZE) {
            double bx, bz, bdir;
            gen_maze(a, env, set, r, ws, lane, bx, bz, bdir, ax, az, adir);
            if (lane != 0) return;
            gen_store_box(a, env, 0, bx, bz, bdir, size, col);
        } else {
            const double brad = sqrt(size * size + size * size) / 2.0;
            double bx, bz;
            gen_place(a, env, set, r, brad, 0, a.gen_args[4], a.gen_args[1], bx, bz);
            const double bdir = rng_uniform(r, -kGenPi, kGenPi);
            // written first so that the agent's placement sees it; colour bias applied below
            gen_store_box(a, env, 0, bx, bz, bdir, size, col);
            // Hallway passes dir=np_random.uniform(-pi/4, pi/4) to place_agent: drawn before the placement
            // (hallway.py:62-65); OneRoom's place_agent() draws the direction after it (miniworld.py:899).
            // The Philox stream keeps its historical order (direction first) for both.
            const bool dir_last = rng_is_pcg(r) && a.generator == MW_GEN_ONEROOM;
            if (!dir_last) adir = rng_uniform(r, -a.gen_args[6], a.gen_args[6]);
            gen_place(a, env, set, r, a.agent_radius, 1, a.gen_args[0], a.gen_args[5], ax, az);
            if (dir_last) adir = rng_uniform(r, -a.gen_args[6], a.gen_args[6]);
        }
        // per-episode parameters (miniworld.py:576-578), then entity randomisation (:584-585)
        for (int k = 0; k < 3; ++k) a.light[(size_t)(0 + k) * N + env] = gen_param(r, a.sky[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(3 + k) * N + env] = gen_param(r, a.light_pos[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(6 + k) * N + env] = gen_param(r, a.light_color[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(9 + k) * N + env] = gen_param(r, a.light_ambient[k], dr);
        for (int k = 0; k < 3; ++k) {
            const double v = col[k] + gen_param(r, a.color_bias[k], dr);      // Box.randomize entity.py:405-407
            a.egeom[((size_t)(3 + k) * a.E + 0) * N + env] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        a.cam[(size_t)0 * N + env] = gen_param(r, a.cam_height, dr);          // Agent.randomize entity.py:505-515
        a.cam[(size_t)1 * N + env] = gen_param(r, a.cam_fwd_disp, dr);
        a.cam[(size_t)2 * N + env] = gen_param(r, a.cam_pitch, dr);
        a.cam[(size_t)3 * N + env] = gen_param(r, a.cam_fov_y, dr);
    }
    if (a.generator == MW_GEN_PROGRAM) {
        gen_program(a, env, set, r, ax, az, adir);
        const bool dr_ = dr;
        for (int k = 0; k < 3; ++k) a.light[(size_t)(0 + k) * N + env] = gen_param(r, a.sky[k], dr_);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(3 + k) * N + env] = gen_param(r, a.light_pos[k], dr_);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(6 + k) * N + env] = gen_param(r, a.light_color[k], dr_);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(9 + k) * N + env] = gen_param(r, a.light_ambient[k], dr_);
        for (int s = 0; s < a.E; ++s)               // Box.randomize in entity order: colour bias (entity.py:405-407)
            if (a.ekind[(size_t)s * N + env] == MW_ENT_BOX)
                for (int k = 0; k < 3; ++k) {
                    const size_t gi = ((size_t)(3 + k) * a.E + s) * N + env;
                    const double v = a.egeom[gi] + gen_param(r, a.color_bias[k], dr_);
                    a.egeom[gi] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                }
        a.cam[(size_t)0 * N + env] = gen_param(r, a.cam_height, dr_);          // Agent.randomize entity.py:505-515
        a.cam[(size_t)1 * N + env] = gen_param(r, a.cam_fwd_disp, dr_);
        a.cam[(size_t)2 * N + env] = gen_param(r, a.cam_pitch, dr_);
        a.cam[(size_t)3 * N + env] = gen_param(r, a.cam_fov_y, dr_);
    }
    if (a.generator == MW_GEN_PICKUP) {
        // pickupobjects.py:55-81: num_objs objects of random kind (Ball, Box, Key) and colour, placed
        // one after the other, then the agent; gen_tab holds the per-kind constants computed by the
        // host from the meshes (radius, height, scale) and the first mesh id of each kind.
        const int n = a.num_objs;
        for (int s = 0; s < n && s < a.E; ++s) {
            const int kind = (int)rng_below(r, 3);          // 0 ball, 1 box, 2 key (obj_types order)
            const int color = (int)rng_below(r, 6);         // index into the sorted COLOR_NAMES
            if (tex_late && s == 0) pick_textures();
            const double radius = a.gt->gen_tab[kind * 4 + 0], height = a.gt->gen_tab[kind * 4 + 1];
            double x, z;
            gen_place(a, env, set, r, radius, s, a.gen_args[0], a.gen_args[1], x, z);
            const double dir = rng_uniform(r, -kGenPi, kGenPi);
            const size_t E = a.E;
            a.ekind[(size_t)s * N + env] = kind == 1 ? MW_ENT_BOX : MW_ENT_MESH;
            a.emesh[(size_t)s * N + env] = kind == 1 ? -1 : (int)a.gt->gen_tab[kind * 4 + 3] + color;
            a.estatic[(size_t)s * N + env] = 0;
            a.epos[((size_t)0 * E + s) * N + env] = x;
            a.epos[((size_t)1 * E + s) * N + env] = 0.0;
            a.epos[((size_t)2 * E + s) * N + env] = z;
            a.edir[(size_t)s * N + env] = dir;
            const double size = kind == 1 ? 0.9 : 0.0;
            for (int k = 0; k < 3; ++k) a.egeom[((size_t)k * E + s) * N + env] = size;
            for (int k = 0; k < 3; ++k) a.egeom[((size_t)(3 + k) * E + s) * N + env] = a.gt->gen_colors[color * 3 + k];
            a.egeom[((size_t)6 * E + s) * N + env] = a.gt->gen_tab[kind * 4 + 2];
            a.egeom[((size_t)7 * E + s) * N + env] = radius;
            a.egeom[((size_t)8 * E + s) * N + env] = height;
        }
        gen_place(a, env, set, r, a.agent_radius, n < a.E ? n : a.E, a.gen_args[0], a.gen_args[1], ax, az);
        adir = rng_uniform(r, -kGenPi, kGenPi);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(0 + k) * N + env] = gen_param(r, a.sky[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(3 + k) * N + env] = gen_param(r, a.light_pos[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(6 + k) * N + env] = gen_param(r, a.light_color[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(9 + k) * N + env] = gen_param(r, a.light_ambient[k], dr);
        for (int s = 0; s < n && s < a.E; ++s)               // Box.randomize: colour bias (entity.py:405-407)
            if (a.ekind[(size_t)s * N + env] == MW_ENT_BOX)
                for (int k = 0; k < 3; ++k) {
                    const size_t gi = ((size_t)(3 + k) * a.E + s) * N + env;
                    const double v = a.egeom[gi] + gen_param(r, a.color_bias[k], dr);
                    a.egeom[gi] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                }
        a.cam[(size_t)0 * N + env] = gen_param(r, a.cam_height, dr);
        a.cam[(size_t)1 * N + env] = gen_param(r, a.cam_fwd_disp, dr);
        a.cam[(size_t)2 * N + env] = gen_param(r, a.cam_pitch, dr);
        a.cam[(size_t)3 * N + env] = gen_param(r, a.cam_fov_y, dr);
    }
    a.ax[env] = ax; a.ay[env] = 0.0; a.az[env] = az; a.adir[env] = adir;
    a.carry[env] = -1; a.step[env] = 0; a.picked[env] = 0;
    if (a.health) a.health[env] = 100;          // CollectHealth._gen_world (collecthealth.py:77)
    if (a.generator == MW_GEN_PROGRAM) {
        for (int k = 0; k < 4; ++k) a.extent[(size_t)k * N + env] = a.prog->p.extent[k];
    } else if (a.generator == MW_GEN_MAZE) {
        const double pitch = a.gt->gen_tab[2] + a.gt->gen_tab[3];
        a.extent[(size_t)0 * N + env] = 0.0; a.extent[(size_t)1 * N + env] = ((int)a.gt->gen_tab[1] - 1) * pitch + a.gt->gen_tab[2];
        a.extent[(size_t)2 * N + env] = 0.0; a.extent[(size_t)3 * N + env] = ((int)a.gt->gen_tab[0] - 1) * pitch + a.gt->gen_tab[2];
    } else {
        for (int k = 0; k < 4; ++k) a.extent[(size_t)k * N + env] = a.gen_args[k];
    }
    rng_store(a.rng, a.N, env, r);
}

// What `info` holds when an episode ends (collecthealth.py:100 health; tmaze.py:89 / ymaze.py:125 goal_pos = the box's position): kept
// before the same-step auto-reset installs the next world (mw_get_final_info; one lane of the env).
__device__ inline void keep_final_info(const MwArgs &a, int env)
{
    if (a.final_health && a.health) a.final_health[env] = a.health[env];
    if (a.final_goal && a.goal_ent >= 0 && a.goal_ent < a.E)
        for (int c = 0; c < 3; ++c) a.final_goal[(size_t)c * a.N + env] = a.epos[((size_t)c * a.E + a.goal_ent) * a.N + env];
}

// An episode ends in spare mode: the env's pre-generated world becomes the live one (64 lanes of one wavefront).
__device__ inline void take_spare(const MwArgs &a, int env, int lane)
{
    const MwSpare &sp = *a.spare;
    const size_t N = a.N, E = a.E;
    if (lane == 0) {
        a.ax[env] = sp.ax[env]; a.ay[env] = sp.ay[env]; a.az[env] = sp.az[env]; a.adir[env] = sp.adir[env];
        for (int k = 0; k < 4; ++k) a.cam[(size_t)k * N + env] = sp.cam[(size_t)k * N + env];
        for (int k = 0; k < 12; ++k) a.light[(size_t)k * N + env] = sp.light[(size_t)k * N + env];
        for (int k = 0; k < 4; ++k) a.extent[(size_t)k * N + env] = sp.extent[(size_t)k * N + env];
        a.carry[env] = -1; a.step[env] = 0; a.picked[env] = 0;
        if (a.health) a.health[env] = 100;
    }
    for (int s = lane; s < (int)E; s += 64) {
        a.ekind[(size_t)s * N + env] = sp.ekind[(size_t)s * N + env];
        a.emesh[(size_t)s * N + env] = sp.emesh[(size_t)s * N + env];
        a.estatic[(size_t)s * N + env] = sp.estatic[(size_t)s * N + env];
        a.edir[(size_t)s * N + env] = sp.edir[(size_t)s * N + env];
        for (int k = 0; k < 3; ++k) a.epos[((size_t)k * E + s) * N + env] = sp.epos[((size_t)k * E + s) * N + env];
        for (int k = 0; k < 9; ++k) a.egeom[((size_t)k * E + s) * N + env] = sp.egeom[((size_t)k * E + s) * N + env];
    }
    if (!a.shared_geom) {
        const int np = sp.npolys[env], ns = sp.nsegs[env];
        const float4 *ps = reinterpret_cast<const float4 *>(sp.polys + (size_t)env * a.max_polys);
        float4 *pd = reinterpret_cast<float4 *>(const_cast<mw_poly *>(a.polys) + (size_t)env * a.max_polys);
        for (int i = lane; i < np * (int)(sizeof(mw_poly) / 16); i += 64) pd[i] = ps[i];
        const double *ss = sp.segs + (size_t)env * a.max_segs * 4;
        double *sd = const_cast<double *>(a.segs) + (size_t)env * a.max_segs * 4;
        for (int i = lane; i < ns * 4; i += 64) sd[i] = ss[i];
        if (lane == 0) { const_cast<int32_t *>(a.npolys)[env] = np; const_cast<int32_t *>(a.nsegs)[env] = ns; if (a.occ_valid) a.occ_valid[env] = 0; }
    }
}

// The same by ONE lane (mw_setup_dense.hip: the env's leading lane; a wavefront holds several envs there).
// Every group of values is loaded into registers first and stored afterwards: source and destination may alias as
// far as the compiler knows, and a load -> store -> load chain would pay one memory round trip per value.
__device__ inline void take_spare_lane(const MwArgs &a, int env)
{
    const MwSpare &sp = *a.spare;
    const size_t N = a.N, E = a.E;
    {
        double v[24];
        v[0] = sp.ax[env]; v[1] = sp.ay[env]; v[2] = sp.az[env]; v[3] = sp.adir[env];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 + k] = sp.cam[(size_t)k * N + env];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[8 + k] = sp.light[(size_t)k * N + env];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[20 + k] = sp.extent[(size_t)k * N + env];
        a.ax[env] = v[0]; a.ay[env] = v[1]; a.az[env] = v[2]; a.adir[env] = v[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.cam[(size_t)k * N + env] = v[4 + k];
#pragma unroll
        for (int k = 0; k < 12; ++k) a.light[(size_t)k * N + env] = v[8 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.extent[(size_t)k * N + env] = v[20 + k];
        a.carry[env] = -1; a.step[env] = 0; a.picked[env] = 0;
    }
    for (int s = 0; s < (int)E; ++s) {
        const int32_t kind = sp.ekind[(size_t)s * N + env], mesh = sp.emesh[(size_t)s * N + env], stat = sp.estatic[(size_t)s * N + env];
        double v[13];
        v[0] = sp.edir[(size_t)s * N + env];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[1 + k] = sp.epos[((size_t)k * E + s) * N + env];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[4 + k] = sp.egeom[((size_t)k * E + s) * N + env];
        a.ekind[(size_t)s * N + env] = kind; a.emesh[(size_t)s * N + env] = mesh; a.estatic[(size_t)s * N + env] = stat;
        a.edir[(size_t)s * N + env] = v[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) a.epos[((size_t)k * E + s) * N + env] = v[1 + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) a.egeom[((size_t)k * E + s) * N + env] = v[4 + k];
    }
    if (!a.shared_geom) {
        const int np = sp.npolys[env], ns = sp.nsegs[env];
        const float4 *ps = reinterpret_cast<const float4 *>(sp.polys + (size_t)env * a.max_polys);
        float4 *pd = reinterpret_cast<float4 *>(const_cast<mw_poly *>(a.polys) + (size_t)env * a.max_polys);
        for (int i = 0; i < np; ++i) {
            float4 q[sizeof(mw_poly) / 16];
#pragma unroll
            for (int k = 0; k < (int)(sizeof(mw_poly) / 16); ++k) q[k] = ps[i * (int)(sizeof(mw_poly) / 16) + k];
#pragma unroll
            for (int k = 0; k < (int)(sizeof(mw_poly) / 16); ++k) pd[i * (int)(sizeof(mw_poly) / 16) + k] = q[k];
        }
        const double *ss = sp.segs + (size_t)env * a.max_segs * 4;
        double *sd = const_cast<double *>(a.segs) + (size_t)env * a.max_segs * 4;
        for (int i = 0; i < ns; ++i) {
            const double s0 = ss[i * 4], s1 = ss[i * 4 + 1], s2 = ss[i * 4 + 2], s3 = ss[i * 4 + 3];
            sd[i * 4] = s0; sd[i * 4 + 1] = s1; sd[i * 4 + 2] = s2; sd[i * 4 + 3] = s3;
        }
        const_cast<int32_t *>(a.npolys)[env] = np; const_cast<int32_t *>(a.nsegs)[env] = ns;
        if (a.occ_valid) a.occ_valid[env] = 0;
    }
}

// Spare refill, executed by extra blocks appended to K1's grid (and by mw_refill_kernel when mw_reset needs the
// spares current): block r handles 64 envs, one per thread — or one env with all 64 lanes for the Maze generator.
// An env is claimed with a compare-and-swap on its refill state, so that its own K1 block, should the new episode
// end at once, either regenerates inline (claim won: state 3) or waits for this block (state 2).
__device__ inline void refill_spares(const MwArgs &a, int r, int tid, unsigned char *ws)
{
    if (tid >= 64) return;
    const bool wpe = a.generator == MW_GEN_MAZE;
    const int env = wpe ? r : r * 64 + tid, lane = wpe ? tid : 0;
    int got = 0;
    if (env < a.N && lane == 0) got = atomicCAS(a.refill_mask + env, 1u, 2u) == 1u;
    if (wpe) got = __shfl(got, 0);
    if (!got) return;
    __threadfence();        // acquire: the live state this world is generated from / over is read behind the claim
    generate_world(*a.gen_spare, env, ws, lane);       // the workspace is the Maze generator's (one env per block there)
    __threadfence();
    if (lane == 0) atomicExch(a.refill_mask + env, 0u);
}

}  // namespace mw
