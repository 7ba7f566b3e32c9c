"""ctypes binding of libmwengine.so (include/mwengine.h) — the only way the Python package
reaches the GPU.  There is deliberately NO CPU fallback: if the HIP library is missing or no
MI355X is visible, construction fails loudly.

The binding passes raw device pointers (``tensor.data_ptr()``) and the current HIP stream;
torch is used for memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# (MW_ENGINE_LIB: a variant build of the library for A/B measurements — tools/perf; the product is csrc/libmwengine.so)
LIB_PATH = os.environ.get("MW_ENGINE_LIB") or os.path.join(_CSRC, "libmwengine.so")

ABI_VERSION = 4
ENT_NONE, ENT_BOX, ENT_MESH, ENT_FRAME = 0, 1, 2, 3
POLY_ENTITY = 0x100          # mw_poly.nv flag: quad of a static entity, not a room
POLY_XF = 0x200              # ... drawn under its own glTranslatef / glRotatef (mw_poly.xf)
POLY_QUAD = 0x400            # ... issued inside glBegin(GL_QUADS) (walls, frames); otherwise GL_POLYGON
TASK_NONE, TASK_GOTO, TASK_PICKUP, TASK_PUTNEXT, TASK_SIDEWALK, TASK_SIGN, TASK_COLLECT = 0, 1, 2, 3, 4, 5, 6
GEN_NONE, GEN_HALLWAY, GEN_ONEROOM, GEN_PICKUP, GEN_MAZE, GEN_PROGRAM = 0, 1, 2, 3, 4, 5
OP_COIN, OP_DRAW_DIR, OP_PLACE, OP_FIXED, OP_BOX_SIZE, OP_COLOR, OP_APPEND = 1, 2, 3, 4, 5, 6, 7
PROG_MAX_ROOMS, PROG_MAX_TEX, PROG_MAX_OPS, PROG_MAX_ENTS = 16, 8, 48, 64
AUTORESET_OFF, AUTORESET_SAME_STEP = 0, 1
OBS_HWC_U8, OBS_CWH_U8, OBS_GREY_F64 = 0, 1, 2
RNG_PHILOX, RNG_PCG64 = 0, 1
PATH_TILE, PATH_QUAD, PATH_QUAD_MESH, PATH_GENERIC = 0, 1, 2, 3

EXPORTS = [
    "mw_create", "mw_destroy", "mw_last_error", "mw_upload_texture", "mw_upload_mesh",
    "mw_set_geometry", "mw_get_geometry", "mw_set_state", "mw_get_state", "mw_set_step_params", "mw_reset",
    "mw_step", "mw_render", "mw_render_top", "mw_render_view", "mw_visible_ents", "mw_set_obs_layout", "mw_pcg64_draws", "mw_check", "mw_kernel_time_ms", "mw_raster_path", "mw_get_info", "mw_get_final_info", "mw_get_list_lengths", "mw_debug_set_mesh_frame_seq", "mw_debug_get_slow_heads",
    "mw_set_gen_program", "mw_selftest_rcp", "mw_selftest_div", "mw_selftest_sort", "mw_selftest_q",
]


class MwRange(C.Structure):
    _fields_ = [("default", C.c_double), ("lo", C.c_double), ("hi", C.c_double)]

    @classmethod
    def of(cls, default, lo=None, hi=None):
        return cls(float(default), float(default if lo is None else lo), float(default if hi is None else hi))


class MwConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device_id", C.c_int32), ("num_envs", C.c_int32),
        ("obs_width", C.c_int32), ("obs_height", C.c_int32), ("msaa", C.c_int32),
        ("max_ents", C.c_int32), ("max_polys", C.c_int32), ("max_segs", C.c_int32),
        ("max_visible", C.c_int32), ("shared_geometry", C.c_int32), ("task", C.c_int32),
        ("goal_ent", C.c_int32), ("goal_ent2", C.c_int32), ("num_objs", C.c_int32), ("max_episode_steps", C.c_int32),
        ("domain_rand", C.c_int32), ("generator", C.c_int32), ("autoreset", C.c_int32),
        ("agent_radius", C.c_double), ("agent_height", C.c_double), ("max_forward_step", C.c_double),
        ("forward_step", MwRange), ("forward_drift", MwRange), ("turn_step", MwRange),
        ("sky_color", MwRange * 3), ("light_pos", MwRange * 3), ("light_color", MwRange * 3),
        ("light_ambient", MwRange * 3), ("obj_color_bias", MwRange * 3),
        ("cam_height", MwRange), ("cam_fwd_disp", MwRange), ("cam_pitch", MwRange), ("cam_fov_y", MwRange),
        ("gen_args", C.c_double * 8),
        ("gen_tab", C.c_double * 12),
        ("gen_colors", C.c_double * 18),
        ("tex_nvar", C.c_int32 * 3), ("tex_var_id", (C.c_int32 * 9) * 3),
        ("tex_var_scale", ((C.c_double * 2) * 9) * 3),
        ("room_wall_height", C.c_double), ("room_no_ceiling", C.c_int32), ("rng_mode", C.c_int32),
    ]


class MwPoly(C.Structure):
    _fields_ = [("v", C.c_float * 12), ("uv", C.c_float * 8), ("n", C.c_float * 3),
                ("nv", C.c_int32), ("tex", C.c_int32), ("rgb", C.c_float * 3), ("xf", C.c_float * 4)]


POLY_DTYPE = np.dtype([("v", np.float32, (4, 3)), ("uv", np.float32, (4, 2)), ("n", np.float32, (3,)),
                       ("nv", np.int32), ("tex", np.int32), ("rgb", np.float32, (3,)), ("xf", np.float32, (4,))])
assert POLY_DTYPE.itemsize == C.sizeof(MwPoly) == 128


class MwProgRoom(C.Structure):
    _fields_ = [("nverts", C.c_int32), ("wall_tex", C.c_int32), ("floor_tex", C.c_int32), ("ceil_tex", C.c_int32),
                ("ox", C.c_double * 4), ("oz", C.c_double * 4), ("nx", C.c_double * 4), ("nz", C.c_double * 4),
                ("min_x", C.c_double), ("max_x", C.c_double), ("min_z", C.c_double), ("max_z", C.c_double), ("cdf", C.c_double)]


class MwProgOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("slot", C.c_int32), ("room", C.c_int32), ("cond", C.c_int32), ("dir_mode", C.c_int32),
                ("flags", C.c_int32), ("lx", C.c_double), ("hx", C.c_double), ("lz", C.c_double), ("hz", C.c_double),
                ("dir", C.c_double), ("a", C.c_double), ("b", C.c_double)]


class MwGenProgram(C.Structure):
    _fields_ = [("n_rooms", C.c_int32), ("n_tex", C.c_int32), ("n_ops", C.c_int32), ("n_ents", C.c_int32),
                ("rooms", MwProgRoom * PROG_MAX_ROOMS),
                ("tex_nvar", C.c_int32 * PROG_MAX_TEX), ("tex_var_id", (C.c_int32 * 9) * PROG_MAX_TEX),
                ("tex_var_scale", ((C.c_double * 2) * 9) * PROG_MAX_TEX),
                ("ops", MwProgOp * PROG_MAX_OPS),
                ("ent_kind", C.c_int32 * PROG_MAX_ENTS), ("ent_mesh", C.c_int32 * PROG_MAX_ENTS), ("ent_static", C.c_int32 * PROG_MAX_ENTS),
                ("ent_pos", (C.c_double * 3) * PROG_MAX_ENTS), ("ent_dir", C.c_double * PROG_MAX_ENTS),
                ("ent_geom", (C.c_double * 9) * PROG_MAX_ENTS),
                ("colors", (C.c_double * 3) * 6), ("extent", C.c_double * 4), ("street", C.c_double * 4),
                ("sign_n", C.c_int32), ("pad", C.c_int32), ("sign_slot", C.c_int32 * 8), ("sign_reward", C.c_double * 8)]


class MwStateView(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "agent_pos", "agent_dir", "cam", "light", "carrying", "step_count", "num_picked_up",
        "ent_kind", "ent_mesh", "ent_static", "ent_pos", "ent_dir", "ent_geom", "extent")]


# name -> (dtype, per-env shape as a function of max_ents)
STATE_FIELDS = {
    "agent_pos": (np.float64, lambda E: (3,)),
    "agent_dir": (np.float64, lambda E: ()),
    "cam": (np.float64, lambda E: (4,)),
    "light": (np.float64, lambda E: (12,)),
    "carrying": (np.int32, lambda E: ()),
    "step_count": (np.int32, lambda E: ()),
    "num_picked_up": (np.int32, lambda E: ()),
    "ent_kind": (np.int32, lambda E: (E,)),
    "ent_mesh": (np.int32, lambda E: (E,)),
    "ent_static": (np.int32, lambda E: (E,)),
    "ent_pos": (np.float64, lambda E: (E, 3)),
    "ent_dir": (np.float64, lambda E: (E,)),
    "ent_geom": (np.float64, lambda E: (E, 9)),
    "extent": (np.float64, lambda E: (4,)),
}


class EngineError(RuntimeError):
    pass


def build_library(force: bool = False) -> str:
    """Compile libmwengine.so for gfx950 with hipcc (in-tree); returns its path."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "mwengine.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        r = subprocess.run([os.path.join(_CSRC, "build.sh")], capture_output=True, text=True)
        if r.returncode != 0:
            raise EngineError("building libmwengine.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libmwengine.so and declare the C ABI.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own HIP runtime (libamdhip64); loading it before our
    # library makes both share ONE runtime instance (same soname), so device pointers and
    # streams can be exchanged.  The other order yields two runtimes in one process.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} not found: the HIP engine is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or miniworld_amd/csrc/build.sh).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    L.mw_create.argtypes = [C.POINTER(MwConfig), C.POINTER(vp)]
    L.mw_destroy.argtypes = [vp]
    L.mw_destroy.restype = None
    L.mw_last_error.argtypes = [vp]
    L.mw_last_error.restype = C.c_char_p
    L.mw_upload_texture.argtypes = [vp, i32, vp, i32, i32]
    L.mw_upload_mesh.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32]
    L.mw_set_geometry.argtypes = [vp, i32, vp, i32, vp, i32]
    L.mw_get_geometry.argtypes = [vp, i32, vp, C.POINTER(i32), vp, C.POINTER(i32)]
    L.mw_set_state.argtypes = [vp, i32, i32, C.POINTER(MwStateView)]
    L.mw_get_state.argtypes = [vp, i32, i32, C.POINTER(MwStateView)]
    L.mw_set_step_params.argtypes = [vp, vp]
    L.mw_set_gen_program.argtypes = [vp, C.POINTER(MwGenProgram), vp, vp, vp, vp, i32, vp, i32]
    L.mw_reset.argtypes = [vp, vp, vp, vp]
    L.mw_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.mw_render.argtypes = [vp, vp, vp, vp]
    L.mw_render_top.argtypes = [vp, vp, vp, i32, vp]
    L.mw_render_view.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.mw_visible_ents.argtypes = [vp, i32, i32, vp, vp]
    L.mw_set_obs_layout.argtypes = [vp, i32]
    L.mw_pcg64_draws.argtypes = [C.c_uint64, i32, vp, vp]
    L.mw_check.argtypes = [vp, vp]
    L.mw_kernel_time_ms.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.mw_raster_path.argtypes = [vp]
    L.mw_get_info.argtypes = [vp, vp, vp, i32, vp]
    L.mw_get_final_info.argtypes = [vp, vp, vp, vp]
    L.mw_get_list_lengths.argtypes = [vp, i32, i32, vp, vp]
    L.mw_debug_set_mesh_frame_seq.argtypes = [vp, C.c_uint32]
    L.mw_debug_get_slow_heads.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _stream_ptr(device=None):
    """torch's current stream ON THE ENGINE'S DEVICE (not on torch's current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """One mw_engine: N environments resident on one GPU."""

    def __init__(self, cfg: MwConfig):
        import torch
        if not torch.cuda.is_available():
            raise EngineError("no ROCm device visible: the mwengine HIP path cannot run (no CPU fallback exists)")
        self.lib = load_library()
        cfg.abi_version = ABI_VERSION
        self.cfg = cfg
        self.N = cfg.num_envs
        self.E = max(cfg.max_ents, 1)
        self.W, self.H = cfg.obs_width, cfg.obs_height
        self.obs_layout = OBS_HWC_U8
        self.device = torch.device("cuda", cfg.device_id)
        h = C.c_void_p()
        rc = self.lib.mw_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(f"mw_create failed ({rc}): {self.lib.mw_last_error(None).decode()}")
        self.h = h

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.mw_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.mw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- assets / world -------------------------------------------------------------
    def upload_texture(self, tex_id: int, rgb_bottom_up: np.ndarray):
        a = np.ascontiguousarray(rgb_bottom_up, np.uint8)
        assert a.ndim == 3 and a.shape[2] == 3
        self._check(self.lib.mw_upload_texture(self.h, tex_id, a.ctypes.data, a.shape[1], a.shape[0]), "mw_upload_texture")

    def upload_mesh(self, mesh_id: int, verts, norms, texcs, colors, tex_id: int = -1):
        """Per-face-vertex arrays [ntris][3][k] as built by miniworld_amd.objmesh.ObjMesh."""
        arrs = [np.ascontiguousarray(a, np.float32) for a in (verts, norms, texcs, colors)]
        n = arrs[0].shape[0]
        assert arrs[0].shape == (n, 3, 3) and arrs[1].shape == (n, 3, 3) and arrs[3].shape == (n, 3, 3)
        self._check(self.lib.mw_upload_mesh(self.h, mesh_id, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                            arrs[2].ctypes.data, arrs[3].ctypes.data, n, tex_id), "mw_upload_mesh")

    def set_geometry(self, env: int, polys: np.ndarray, segs: np.ndarray):
        p = np.ascontiguousarray(polys, POLY_DTYPE)
        s = np.ascontiguousarray(segs, np.float64).reshape(-1, 4)
        self._check(self.lib.mw_set_geometry(self.h, env, p.ctypes.data, len(p), s.ctypes.data, len(s)), "mw_set_geometry")

    def get_geometry(self, env: int):
        polys = np.zeros(self.cfg.max_polys, POLY_DTYPE)
        segs = np.zeros((self.cfg.max_segs, 2, 2), np.float64)
        npoly, nseg = C.c_int32(), C.c_int32()
        self._check(self.lib.mw_get_geometry(self.h, env, polys.ctypes.data, C.byref(npoly), segs.ctypes.data,
                                             C.byref(nseg)), "mw_get_geometry")
        return polys[:npoly.value], segs[:nseg.value]

    def _view(self, arrays: dict, count: int, alloc: bool):
        view, keep = MwStateView(), {}
        for name, (dt, shp) in STATE_FIELDS.items():
            if alloc:
                arr = np.zeros((count,) + shp(self.E), dt)
            elif name in arrays and arrays[name] is not None:
                arr = np.ascontiguousarray(arrays[name], dt).reshape((count,) + shp(self.E))
            else:
                continue
            keep[name] = arr
            setattr(view, name, arr.ctypes.data)
        return view, keep

    def set_state(self, arrays: dict, first: int = 0, count: int | None = None):
        count = self.N - first if count is None else count
        view, keep = self._view(arrays, count, alloc=False)
        self._check(self.lib.mw_set_state(self.h, first, count, C.byref(view)), "mw_set_state")

    def get_state(self, first: int = 0, count: int | None = None) -> dict:
        count = self.N - first if count is None else count
        view, keep = self._view({}, count, alloc=True)
        self._check(self.lib.mw_get_state(self.h, first, count, C.byref(view)), "mw_get_state")
        return keep

    def set_gen_program(self, prog: MwGenProgram, polys: np.ndarray, poly_room, poly_surf, poly_m, segs: np.ndarray):
        """Installs the placement program of an MW_GEN_PROGRAM engine (include/mwengine.h: mw_set_gen_program)."""
        p = np.ascontiguousarray(polys, POLY_DTYPE)
        room = np.ascontiguousarray(poly_room, np.int32)
        surf = np.ascontiguousarray(poly_surf, np.int32)
        m = np.ascontiguousarray(poly_m, np.float64).reshape(len(p), 4, 2)
        sg = np.ascontiguousarray(segs, np.float64).reshape(-1, 4)
        self._check(self.lib.mw_set_gen_program(self.h, C.byref(prog), p.ctypes.data, room.ctypes.data, surf.ctypes.data,
                                                m.ctypes.data, len(p), sg.ctypes.data, len(sg)), "mw_set_gen_program")

    def set_step_params(self, params: np.ndarray | None):
        if params is None:
            self._check(self.lib.mw_set_step_params(self.h, None), "mw_set_step_params")
        else:
            p = np.ascontiguousarray(params, np.float64).reshape(self.N, 3)
            self._check(self.lib.mw_set_step_params(self.h, p.ctypes.data), "mw_set_step_params")

    def reset(self, mask: np.ndarray | None = None, seeds: np.ndarray | None = None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        self._check(self.lib.mw_reset(self.h, None if m is None else m.ctypes.data,
                                      None if s is None else s.ctypes.data, _stream_ptr(self.device)), "mw_reset")

    # -- hot path ---------------------------------------------------------------------
    def _dev_tensor(self, t, name, dtype, numel):
        """The kernels read raw pointers: a tensor of another dtype / device / stride pattern would be read as garbage
        (an int64 action tensor as int32 pairs, a CPU tensor as a fault).  Outputs must already be right; see step()."""
        if t is None:
            return None
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous() or t.numel() != numel:
            raise EngineError(f"{name}: need a contiguous {dtype} tensor of {numel} elements on {self.device}, got "
                              f"{t.dtype} {tuple(t.shape)} on {t.device}{'' if t.is_contiguous() else ' (non-contiguous)'}")
        return t

    def step(self, actions, obs, depth=None, reward=None, term=None, trunc=None):
        """All arguments are torch tensors on this engine's device (depth may be None).  `actions` is converted to a
        contiguous int32 tensor on the device if it is not one already (torch.randint / argmax / Categorical.sample
        give int64; a column of a [N, T] tensor is strided); the output tensors are checked, never converted."""
        import torch
        if actions.device != self.device or actions.dtype != torch.int32 or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.numel() != self.N:
            raise EngineError(f"actions: {actions.numel()} elements for {self.N} envs")
        obs_numel = self.N * self.H * self.W * (1 if self.obs_layout == OBS_GREY_F64 else 3)
        self._dev_tensor(obs, "obs", torch.float64 if self.obs_layout == OBS_GREY_F64 else torch.uint8, obs_numel)
        self._dev_tensor(depth, "depth", torch.float32, self.N * self.H * self.W)
        self._dev_tensor(reward, "reward", torch.float32, self.N)
        self._dev_tensor(term, "terminated", torch.uint8, self.N)
        self._dev_tensor(trunc, "truncated", torch.uint8, self.N)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.mw_step(self.h, ptr(actions), ptr(obs), ptr(depth), ptr(reward), ptr(term),
                                     ptr(trunc), _stream_ptr(self.device)), "mw_step")

    def render(self, obs, depth=None):
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.mw_render(self.h, ptr(obs), ptr(depth), _stream_ptr(self.device)), "mw_render")

    def render_top(self, obs, depth=None, render_agent=True):
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.mw_render_top(self.h, ptr(obs), ptr(depth), int(render_agent), _stream_ptr(self.device)), "mw_render_top")

    def render_view(self, env: int, width: int, height: int, msaa: int = 16, top: bool = False,
                    render_agent: bool = False, want_depth: bool = False):
        """One env into a (height, width) buffer with 8 or 16 samples; returns torch tensors."""
        import torch
        out = torch.zeros((height, width, 3), dtype=torch.uint8, device=self.device)
        dep = torch.zeros((height, width, 1), dtype=torch.float32, device=self.device) if want_depth else None
        flags = (1 if top else 0) | (2 if render_agent else 0)
        self._check(self.lib.mw_render_view(self.h, env, flags, width, height, msaa, C.c_void_p(out.data_ptr()),
                                            None if dep is None else C.c_void_p(dep.data_ptr()), _stream_ptr(self.device)),
                    "mw_render_view")
        return (out, dep) if want_depth else out

    def set_obs_layout(self, layout: int):
        """Layout of the obs buffer the raster kernel writes: OBS_HWC_U8 | OBS_CWH_U8 | OBS_GREY_F64."""
        self._check(self.lib.mw_set_obs_layout(self.h, int(layout)), "mw_set_obs_layout")
        self.obs_layout = int(layout)

    def obs_buffer(self, count: int | None = None):
        """A device tensor of the right shape / dtype for the current obs layout."""
        import torch
        n = self.N if count is None else count
        layout = self.obs_layout
        if layout == OBS_CWH_U8:
            return torch.zeros((n, 3, self.W, self.H), dtype=torch.uint8, device=self.device)
        if layout == OBS_GREY_F64:
            return torch.zeros((n, self.H, self.W, 1), dtype=torch.float64, device=self.device)
        return torch.zeros((n, self.H, self.W, 3), dtype=torch.uint8, device=self.device)

    def visible_ents(self, first_env: int = 0, count: int | None = None):
        """get_visible_ents (miniworld.py:1238-1333): uint8[count, max_ents] on the device, 1 = visible."""
        import torch
        count = self.N - first_env if count is None else count
        vis = torch.zeros((count, self.E), dtype=torch.uint8, device=self.device)
        self._check(self.lib.mw_visible_ents(self.h, first_env, count, C.c_void_p(vis.data_ptr()), _stream_ptr(self.device)),
                    "mw_visible_ents")
        return vis

    def check(self):
        self._check(self.lib.mw_check(self.h, _stream_ptr(self.device)), "mw_check")

    def get_info(self, health=None, ent_pos=None, ent_slot=0):
        """Fills the caller's device tensors: health int32[N] (CollectHealth's info["health"]) and / or ent_pos float64[N, 3]
        (position of entity slot ent_slot: TMaze / YMaze info["goal_pos"])."""
        import torch
        for t, dt, shape in ((health, torch.int32, (self.N,)), (ent_pos, torch.float64, (self.N, 3))):
            if t is not None:
                assert t.is_cuda and t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous()
        self._check(self.lib.mw_get_info(self.h, health.data_ptr() if health is not None else None,
                                         ent_pos.data_ptr() if ent_pos is not None else None, int(ent_slot), _stream_ptr(self.device)), "mw_get_info")

    def get_final_info(self, health=None, goal_pos=None):
        """The `info` values of each env's last FINISHED episode (kept by the step kernel before the same-step auto-reset): health
        int32[N] (CollectHealth) and / or goal_pos float64[N, 3] (TMaze / YMaze)."""
        import torch
        for t, dt, shape in ((health, torch.int32, (self.N,)), (goal_pos, torch.float64, (self.N, 3))):
            if t is not None:
                assert t.is_cuda and t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous()
        self._check(self.lib.mw_get_final_info(self.h, health.data_ptr() if health is not None else None,
                                               goal_pos.data_ptr() if goal_pos is not None else None, _stream_ptr(self.device)), "mw_get_final_info")

    def list_lengths(self):
        """int32[N]: triangles in each env's display list of the last frame (after clipping and culling)."""
        out = np.zeros(self.N, np.int32)
        self._check(self.lib.mw_get_list_lengths(self.h, 0, self.N, out.ctypes.data, _stream_ptr(self.device)), "mw_get_list_lengths")
        return out

    def raster_path(self):
        """Which raster kernels drew the last frame: PATH_TILE / PATH_QUAD / PATH_QUAD_MESH / PATH_GENERIC (mwengine.h)."""
        return int(self.lib.mw_raster_path(self.h))

    def kernel_time_ms(self, reset=0):
        """(raster ms, setup ms, launches measured) since the last call; reset = k > 0: time one launch in k from now
        on (1 = every launch), 0: the default one in 8, < 0: switch timing off."""
        r, s, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.mw_kernel_time_ms(self.h, reset, C.byref(r), C.byref(s), C.byref(n)), "mw_kernel_time_ms")
        return r.value, s.value, n.value


# ------------------------------------------------------------------ DomainParams -> config

def default_ranges() -> dict:
    """The reference's DEFAULT_PARAMS table (params.py:115-130) as (default, min, max)."""
    return {
        "sky_color": ([0.25, 0.82, 1.0], [0.1, 0.1, 0.1], [1.0, 1.0, 1.0]),
        "light_pos": ([0, 2.5, 0], [-40, 2.5, -40], [40, 5, 40]),
        "light_color": ([0.7, 0.7, 0.7], [0.45, 0.45, 0.45], [0.8, 0.8, 0.8]),
        "light_ambient": ([0.45, 0.45, 0.45], [0.35, 0.35, 0.35], [0.55, 0.55, 0.55]),
        "obj_color_bias": ([0, 0, 0], [-0.2, -0.2, -0.2], [0.2, 0.2, 0.2]),
        "forward_step": (0.15, 0.12, 0.17),
        "forward_drift": (0.0, -0.05, 0.05),
        "turn_step": (15.0, 10.0, 20.0),
        "cam_pitch": (0.0, -5.0, 5.0),
        "cam_fov_y": (60.0, 55.0, 65.0),
        "cam_height": (1.5, 1.45, 1.55),
        "cam_fwd_disp": (0.0, -0.05, 0.10),
    }


def fill_ranges(cfg: MwConfig, ranges: dict | None = None):
    r = default_ranges() if ranges is None else ranges
    for name in ("forward_step", "forward_drift", "turn_step", "cam_pitch", "cam_fov_y", "cam_height", "cam_fwd_disp"):
        setattr(cfg, name, MwRange.of(*r[name]))
    for name in ("sky_color", "light_pos", "light_color", "light_ambient", "obj_color_bias"):
        d, lo, hi = r[name]
        arr = getattr(cfg, name)
        for k in range(3):
            arr[k] = MwRange.of(d[k], lo[k], hi[k])
    cfg.max_forward_step = float(r["forward_step"][2])
