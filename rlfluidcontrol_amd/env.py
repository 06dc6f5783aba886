"""ctypes binding of libafc_env.so (include/afc_env.h) and the host-side mirror of the
reference's `AFCCylinder` interface (clientLilypad/AFCCylinder.pde) for a batch of envs.

There is NO CPU fallback: importing works anywhere (so the ABI can be inspected), but
creating an env raises unless the HIP extension is built and a GPU is visible.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libafc_env.so")

AFC_MEM_HOST, AFC_MEM_DEVICE = 0, 1
FIELD_U, FIELD_V, FIELD_P, FIELD_VORT = 0, 1, 2, 3
STATIC = {"del.x": 0, "del.y": 1, "del1.x": 2, "del1.y": 3, "wnx.x": 4, "wnx.y": 5, "wny.x": 6,
          "wny.y": 7, "c.x": 8, "c.y": 9, "w1.x": 10, "w2.x": 11, "w1.y": 12, "w2.y": 13}

# every symbol include/afc_env.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "afc_default_config", "afc_create", "afc_destroy", "afc_last_error", "afc_dims", "afc_num_envs",
    "afc_num_levels", "afc_dt", "afc_set_io_space", "afc_set_stream", "afc_synchronize",
    "afc_load_state_bdim", "afc_set_reset_state", "afc_save_state_bdim", "afc_reset", "afc_sim_step",
    "afc_agent_step", "afc_get_time", "afc_get_solver_stats", "afc_get_diverged", "afc_get_field",
    "afc_set_field", "afc_get_probes", "afc_get_static", "afc_level_dims", "afc_op_advdif",
    "afc_op_mg_solve", "afc_op_setbc_u", "afc_profile_phases", "afc_format_float_java", "afc_get_cfl",
    "afc_get_window_complete", "afc_get_mg_histogram", "afc_measure_copy_rate", "afc_op_project",
    "afc_placement_info", "afc_fused_tail", "afc_upleg",
]


class AfcConfig(C.Structure):
    """struct afc_config (include/afc_env.h); constants of clientCFD.pde:5-14,94-96."""
    _fields_ = [
        ("resolution", C.c_int32), ("x_lengths", C.c_int32), ("y_lengths", C.c_int32),
        ("reynolds", C.c_int32), ("t_step", C.c_float), ("dR", C.c_float), ("gR", C.c_float),
        ("theta", C.c_float), ("act_scale", C.c_float), ("substeps", C.c_int32),
        ("init_time", C.c_float), ("end_time", C.c_float), ("num_envs", C.c_int32),
        ("device", C.c_int32), ("mg_max_iters", C.c_int32), ("carry_quirk", C.c_int32),
        ("reward_penalty", C.c_float), ("exact_sums", C.c_int32), ("impl", C.c_int32), ("stagger_us", C.c_int32), ("reserved", C.c_int32 * 1),
        ("gs_mode", C.c_int32), ("fuse_tail", C.c_int32), ("upleg", C.c_int32),
    ]


class AfcError(RuntimeError):
    pass


_lib = None


def load_library():
    """Loads libafc_env.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:  # PyTorch ships its own HIP runtime under the same SONAME: when both live in one process
        import torch  # noqa: F401  (torch's copy must be the one the loader binds first)
    except ImportError:
        pass
    path = os.environ.get("AFC_ENV_LIB", LIB_PATH)  # override: A/B runs of two builds on one GPU box
    if not os.path.exists(path):
        raise AfcError("libafc_env.so is missing: build it with `python -m rlfluidcontrol_amd.build` "
                       "(the product has no CPU fallback)")
    L = C.CDLL(path)
    vp, i32p, f32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p
    L.afc_last_error.restype = C.c_char_p
    L.afc_default_config.argtypes = [C.POINTER(AfcConfig)]
    L.afc_create.argtypes = [C.POINTER(AfcConfig), C.POINTER(vp)]
    L.afc_destroy.argtypes = [vp]
    L.afc_dims.argtypes = [vp, i32p, i32p]
    L.afc_num_envs.argtypes = [vp]
    L.afc_num_levels.argtypes = [vp]
    L.afc_dt.argtypes = [vp]
    L.afc_dt.restype = C.c_float
    L.afc_set_io_space.argtypes = [vp, C.c_int]
    L.afc_set_stream.argtypes = [vp, vp]
    L.afc_synchronize.argtypes = [vp]
    L.afc_load_state_bdim.argtypes = [vp, C.c_char_p]
    L.afc_set_reset_state.argtypes = [vp, f32p, f32p, f32p]
    L.afc_save_state_bdim.argtypes = [vp, C.c_int, C.c_char_p]
    L.afc_reset.argtypes = [vp, u8p]
    L.afc_sim_step.argtypes = [vp, f32p, f32p]
    L.afc_agent_step.argtypes = [vp, f32p, f32p, f32p, u8p, vp]
    L.afc_get_time.argtypes = [vp, f32p]
    L.afc_get_solver_stats.argtypes = [vp, vp, f32p]
    L.afc_get_diverged.argtypes = [vp, u8p]
    L.afc_get_field.argtypes = [vp, C.c_int, C.c_int, f32p]
    L.afc_set_field.argtypes = [vp, C.c_int, C.c_int, f32p]
    L.afc_get_probes.argtypes = [vp, C.c_int, f32p]
    L.afc_get_static.argtypes = [vp, C.c_int, C.c_int, f32p]
    L.afc_level_dims.argtypes = [vp, C.c_int, i32p, i32p]
    L.afc_op_advdif.argtypes = [vp, f32p, f32p, f32p, f32p]
    L.afc_op_mg_solve.argtypes = [vp, f32p, f32p, i32p, C.POINTER(C.c_float)]
    L.afc_op_setbc_u.argtypes = [vp, f32p, f32p]
    try:  # entry points added in round 4: absent from the older builds tools/ab.sh compares against (AFC_ENV_LIB)
        L.afc_op_project.argtypes = [vp, f32p, f32p, f32p, i32p, C.POINTER(C.c_float)]
        L.afc_measure_copy_rate.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_float)]
    except AttributeError:
        if not os.environ.get("AFC_ENV_LIB"):
            raise
    try:  # round 5
        L.afc_placement_info.argtypes = [vp, i32p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.afc_fused_tail.argtypes = [vp]
        L.afc_upleg.argtypes = [vp]
    except AttributeError:
        if not os.environ.get("AFC_ENV_LIB"):
            raise
    L.afc_profile_phases.argtypes = [vp, C.c_int, vp, C.c_int]
    L.afc_format_float_java.argtypes = [C.c_float, C.c_char_p, C.c_int]
    L.afc_get_cfl.argtypes = [vp, f32p]
    L.afc_get_window_complete.argtypes = [vp, u8p]
    L.afc_get_mg_histogram.argtypes = [vp, vp, C.c_int]
    _lib = L
    return L


def default_config(**overrides):
    L = load_library()
    cfg = AfcConfig()
    L.afc_default_config(C.byref(cfg))
    # development: AFC_DBG=<switches> runs any program (e.g. the test suite) on an experiment path -- but only together
    # with AFC_DBG_ENABLE=1, so that a stray or inherited AFC_DBG cannot silently change physics or benchmark numbers
    if os.environ.get("AFC_DBG"):
        if os.environ.get("AFC_DBG_ENABLE") == "1":
            cfg.reserved[0] = int(os.environ["AFC_DBG"], 0)
            print("rlfluidcontrol_amd: AFC_DBG=%s is ON (development switches; results are not the product's)"
                  % os.environ["AFC_DBG"], file=sys.stderr)
        else:
            print("rlfluidcontrol_amd: AFC_DBG=%s ignored (set AFC_DBG_ENABLE=1 to honour it)" % os.environ["AFC_DBG"],
                  file=sys.stderr)
    for k, v in overrides.items():
        if k == "dbg":  # experiment switches (reserved[0])
            cfg.reserved[0] |= int(v)
            continue
        if not hasattr(cfg, k):
            raise AfcError("unknown config key %r" % k)
        setattr(cfg, k, v)
    return cfg


def java_float_text(x):
    """Float.toString(x) as the C library prints it in .bdim files (afc_format_float_java)."""
    L = load_library()
    buf = C.create_string_buffer(32)
    if L.afc_format_float_java(C.c_float(float(x)), buf, 32) != 0:
        raise AfcError(L.afc_last_error().decode())
    return buf.value.decode()


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class AFCEnv:
    """A batch of `num_envs` independent AFCCylinder simulations on one MI355X.

    Mirrors the members clientCFD.pde uses: update2() (one sim-step), force, t, flow.p,
    resume()/write(), plus the batched agent step (16 sim-steps + averaging).
    Host-numpy I/O by default; `use_device_io()` switches the per-env I/O arrays to raw device
    pointers (torch tensors) so an agent can stay on the GPU.
    """

    def __init__(self, num_envs=1, resolution=24, reynolds=500, device=0, **kw):
        L = load_library()
        self._L = L
        self.cfg = default_config(num_envs=num_envs, resolution=resolution, reynolds=reynolds, device=device, **kw)
        h = C.c_void_p()
        rc = L.afc_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise AfcError("afc_create failed (%d): %s" % (rc, L.afc_last_error().decode()))
        self._h = h
        n, m = C.c_int32(), C.c_int32()
        L.afc_dims(h, C.byref(n), C.byref(m))
        self.n, self.m = n.value, m.value
        self.num_envs = num_envs
        self.resolution = resolution
        self.dt = L.afc_dt(h)
        self.num_levels = L.afc_num_levels(h)
        self._device_io = False

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._L.afc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise AfcError("%s failed (%d): %s" % (what, rc, self._L.afc_last_error().decode()))

    # -- configuration --
    def use_device_io(self, stream_ptr=None):
        self._ck(self._L.afc_set_io_space(self._h, AFC_MEM_DEVICE), "afc_set_io_space")
        self._ck(self._L.afc_set_stream(self._h, C.c_void_p(stream_ptr or 0)), "afc_set_stream")
        self._device_io = True

    def use_host_io(self):
        self._ck(self._L.afc_set_io_space(self._h, AFC_MEM_HOST), "afc_set_io_space")
        self._device_io = False

    def synchronize(self):
        self._ck(self._L.afc_synchronize(self._h), "afc_synchronize")

    # -- state --
    def resume(self, path):
        """BDIM.resume (BDIM.pde:239-251) into every env; becomes the reset state."""
        self._ck(self._L.afc_load_state_bdim(self._h, os.fsencode(path)), "afc_load_state_bdim")

    def write(self, path, env_idx=0):
        """BDIM.write (BDIM.pde:226-237)."""
        self._ck(self._L.afc_save_state_bdim(self._h, env_idx, os.fsencode(path)), "afc_save_state_bdim")

    def set_reset_state(self, u, v, p):
        u, v, p = (_f32(a, (self.n, self.m)) for a in (u, v, p))
        self._ck(self._L.afc_set_reset_state(self._h, _np_ptr(u), _np_ptr(v), _np_ptr(p)), "afc_set_reset_state")

    def reset(self, mask=None):
        if mask is None:
            self._ck(self._L.afc_reset(self._h, None), "afc_reset")
        elif self._device_io:
            self._ck(self._L.afc_reset(self._h, C.c_void_p(mask)), "afc_reset")
        else:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            self._ck(self._L.afc_reset(self._h, _np_ptr(mk)), "afc_reset")

    def get_field(self, which, env_idx=0):
        out = np.empty((self.n, self.m), np.float32)
        self._ck(self._L.afc_get_field(self._h, env_idx, which, _np_ptr(out)), "afc_get_field")
        return out

    def set_field(self, which, arr, env_idx=0):
        a = _f32(arr, (self.n, self.m))
        self._ck(self._L.afc_set_field(self._h, env_idx, which, _np_ptr(a)), "afc_set_field")

    def get_state(self, env_idx=0):
        return tuple(self.get_field(w, env_idx) for w in (FIELD_U, FIELD_V, FIELD_P))

    def vorticity(self, env_idx=0):
        """flow.u.curl() (VectorField.pde:78-87)."""
        return self.get_field(FIELD_VORT, env_idx)

    def probes(self, num=32):
        out = np.empty((self.num_envs, num), np.float32)
        self._ck(self._L.afc_get_probes(self._h, num, _np_ptr(out)), "afc_get_probes")
        return out

    def static(self, name, level=0):
        n, m = C.c_int32(), C.c_int32()
        self._ck(self._L.afc_level_dims(self._h, level, C.byref(n), C.byref(m)), "afc_level_dims")
        out = np.empty((n.value, m.value), np.float32)
        self._ck(self._L.afc_get_static(self._h, STATIC[name], level, _np_ptr(out)), "afc_get_static")
        return out

    @property
    def t(self):
        out = np.empty(self.num_envs, np.float32)
        self._ck(self._L.afc_get_time(self._h, _np_ptr(out)), "afc_get_time")
        return out

    def solver_stats(self):
        it = np.empty((self.num_envs, 2), np.int32)
        rs = np.empty((self.num_envs, 2), np.float32)
        self._ck(self._L.afc_get_solver_stats(self._h, _np_ptr(it), _np_ptr(rs)), "afc_get_solver_stats")
        return it, rs

    def cfl(self):
        """BDIM.checkCFL without the min(.,1) (VectorField.pde:225-235) per env."""
        out = np.empty(self.num_envs, np.float32)
        self._ck(self._L.afc_get_cfl(self._h, _np_ptr(out)), "afc_get_cfl")
        return out

    def window_complete(self):
        out = np.empty(self.num_envs, np.uint8)
        self._ck(self._L.afc_get_window_complete(self._h, _np_ptr(out)), "afc_get_window_complete")
        return out

    def mg_histogram(self, clear=False):
        """solves that took 1..7, >=8 multigrid iterations since the last clear (all envs)"""
        out = np.zeros(8, np.int64)
        self._ck(self._L.afc_get_mg_histogram(self._h, _np_ptr(out), int(clear)), "afc_get_mg_histogram")
        return out

    def placement_info(self):
        """what afc_create's placement search did: {"trials", "best_GBs", "median_GBs", "hit"} (include/afc_env.h)"""
        if not hasattr(self._L, "afc_placement_info"):
            return None
        t, b, m = C.c_int32(0), C.c_float(0), C.c_float(0)
        self._ck(self._L.afc_placement_info(self._h, C.byref(t), C.byref(b), C.byref(m)), "afc_placement_info")
        return {"trials": int(t.value), "best_GBs": round(float(b.value), 1), "median_GBs": round(float(m.value), 1),
                "hit": bool(t.value > 1 and b.value >= 1.035 * m.value)}

    def fused_tail(self):
        """True if the step kernels run the predictor's projection tail inside the corrector's advection (afc_config.fuse_tail)"""
        if not hasattr(self._L, "afc_fused_tail"):
            return None
        r = self._L.afc_fused_tail(self._h)
        if r < 0:
            self._ck(r, "afc_fused_tail")
        return bool(r)

    def upleg(self):
        """the up-leg the step kernels run (afc_config.upleg resolved): 1 strip pipeline, 2 barrier form, 3 marching down the rows"""
        r = self._L.afc_upleg(self._h)
        if r < 0:
            self._ck(r, "afc_upleg")
        return int(r)

    def diverged(self):
        out = np.empty(self.num_envs, np.uint8)
        self._ck(self._L.afc_get_diverged(self._h, _np_ptr(out)), "afc_get_diverged")
        return out

    # -- stepping --
    def update2(self, xi_m):
        """AFCCylinder.update2 for every env. xi_m [E,2] -> force [E,2]."""
        xi = _f32(xi_m, (self.num_envs, 2))
        force = np.empty((self.num_envs, 2), np.float32)
        self._ck(self._L.afc_sim_step(self._h, _np_ptr(xi), _np_ptr(force)), "afc_sim_step")
        return force

    def agent_step(self, action):
        """action [E,2] -> (obs [E,2]=(Cl,Cd), reward [E], done [E], nsteps [E])."""
        a = _f32(action, (self.num_envs, 2))
        obs = np.empty((self.num_envs, 2), np.float32)
        rew = np.empty(self.num_envs, np.float32)
        done = np.empty(self.num_envs, np.uint8)
        ns = np.empty(self.num_envs, np.int32)
        self._ck(self._L.afc_agent_step(self._h, _np_ptr(a), _np_ptr(obs), _np_ptr(rew), _np_ptr(done), _np_ptr(ns)),
                 "afc_agent_step")
        return obs, rew, done, ns

    def sim_step_device(self, xi_ptr, force_ptr):
        """raw device pointers (async on the handle's stream)"""
        self._ck(self._L.afc_sim_step(self._h, C.c_void_p(xi_ptr), C.c_void_p(force_ptr)), "afc_sim_step")

    def agent_step_device(self, action_ptr, obs_ptr, reward_ptr, done_ptr, nsteps_ptr=0):
        self._ck(self._L.afc_agent_step(self._h, C.c_void_p(action_ptr), C.c_void_p(obs_ptr), C.c_void_p(reward_ptr),
                                        C.c_void_p(done_ptr), C.c_void_p(nsteps_ptr or 0)), "afc_agent_step")

    PHASES = ["advdif", "forcing", "setbc", "resid", "down0", "coarse_down", "coarse_gs", "coarse_up", "up0",
              "gs0", "inc0", "rr", "project", "heun"]

    def profile_phases(self, enable=True):
        """returns {phase: microseconds} accumulated by env 0 since the last call"""
        out = np.zeros(32, np.int64)
        self._ck(self._L.afc_profile_phases(self._h, int(enable), _np_ptr(out), 32), "afc_profile_phases")
        d = {k: out[i] / 100.0 for i, k in enumerate(self.PHASES)}
        for l in range(1, 8):
            if out[16 + l]:
                d["up_L%d" % l] = out[16 + l] / 100.0
            if out[24 + l]:
                d["down_L%d" % l] = out[24 + l] / 100.0
        return d

    # -- unit-level operators (parity tests) --
    def op_advdif(self, fx, fy, u0x=None, u0y=None):
        """VectorField.AdvDif; u0x/u0y None: u0 = the advected field itself (BDIM.update, the predictor's form)"""
        fx, fy = (_f32(a, (self.n, self.m)).copy() for a in (fx, fy))
        if u0x is None:
            u0x, u0y = fx, fy
        else:
            u0x, u0y = (_f32(a, (self.n, self.m)).copy() for a in (u0x, u0y))
        self._ck(self._L.afc_op_advdif(self._h, _np_ptr(fx), _np_ptr(fy), _np_ptr(u0x), _np_ptr(u0y)), "afc_op_advdif")
        return fx, fy

    def op_project(self, ux, uy, p):
        """VectorField.project(c, p): returns (ux, uy, p, iterations, L_inf residual)"""
        ux, uy, p = (_f32(a, (self.n, self.m)).copy() for a in (ux, uy, p))
        it = C.c_int32()
        linf = C.c_float()
        self._ck(self._L.afc_op_project(self._h, _np_ptr(ux), _np_ptr(uy), _np_ptr(p), C.byref(it), C.byref(linf)),
                 "afc_op_project")
        return ux, uy, p, it.value, linf.value

    def op_setbc_u(self, ux, uy):
        ux, uy = (_f32(a, (self.n, self.m)).copy() for a in (ux, uy))
        self._ck(self._L.afc_op_setbc_u(self._h, _np_ptr(ux), _np_ptr(uy)), "afc_op_setbc_u")
        return ux, uy

    def op_mg_solve(self, x, b):
        x, b = (_f32(a, (self.n, self.m)).copy() for a in (x, b))
        it = C.c_int32()
        linf = C.c_float()
        self._ck(self._L.afc_op_mg_solve(self._h, _np_ptr(x), _np_ptr(b), C.byref(it), C.byref(linf)), "afc_op_mg_solve")
        return x, it.value, linf.value
