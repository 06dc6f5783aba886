"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rlfluidcontrol_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "afc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    for name in ("orc_n", "orc_m"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("orc_dt", "orc_nu", "orc_t", "orc_flow_t"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_set_t.argtypes = [C.c_void_p, C.c_float]
    L.orc_set_state.argtypes = [C.c_void_p, f32p, f32p, f32p]
    L.orc_get_state.argtypes = [C.c_void_p, f32p, f32p, f32p]
    L.orc_resume.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_resume.restype = C.c_int
    L.orc_write.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_write.restype = C.c_int
    L.orc_update2.argtypes = [C.c_void_p, C.c_float, C.c_float, f32p, C.POINTER(C.c_int), f32p]
    L.orc_get_coeff.argtypes = [C.c_void_p, C.c_int, f32p]
    L.orc_refresh_coeffs.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.orc_weights.argtypes = [C.c_void_p, C.c_float, C.c_float, f32p]
    L.orc_cfl.argtypes = [C.c_void_p]
    L.orc_cfl.restype = C.c_float
    L.orc_vorticity.argtypes = [C.c_void_p, f32p]
    L.orc_probes.argtypes = [C.c_void_p, C.c_int, f32p]
    L.orc_force.argtypes = [C.c_void_p, f32p]
    L.orc_poisson_times.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_poisson_diag_inv.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_mg_updates.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_int, f32p]
    L.orc_mg_updates.restype = C.c_float
    L.orc_mg_solve.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_int, C.POINTER(C.c_float)]
    L.orc_mg_solve.restype = C.c_int
    L.orc_mg_smooth.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_int]
    L.orc_setbc.argtypes = [C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_int]
    L.orc_advdif.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_float, C.c_float]
    L.orc_polygon_distance.argtypes = [C.c_int, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_polygon_distance.restype = C.c_float
    L.orc_linear.argtypes = [C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float]
    L.orc_linear.restype = C.c_float
    for name in ("orc_delta0", "orc_delta1", "orc_union_delta0"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    _LIB = L
    return L


COEFF_NAMES = ["del.x", "del.y", "del1.x", "del1.y", "wnx.x", "wnx.y", "wny.x", "wny.y",
               "ub.x", "ub.y", "distance.x", "distance.y", "c.x", "c.y"]


class Oracle:
    """One env of the reference's AFCCylinder (AFCCylinder.pde:1-61), CPU float32."""

    def __init__(self, resolution=24, Re=500, dR=0.125, gR=0.2, theta=None, tStep=0.0075,
                 xLengths=16, yLengths=8):
        L = lib()
        if theta is None:
            theta = float(np.float32(np.float32(np.pi) / np.float32(3)))  # PI/3 in float
        self._L = L
        self._h = L.orc_create(resolution, Re, dR, gR, theta, tStep, xLengths, yLengths)
        self.n = L.orc_n(self._h)
        self.m = L.orc_m(self._h)
        self.resolution = resolution
        self.dt = L.orc_dt(self._h)
        self.nu = L.orc_nu(self._h)

    def __del__(self):
        try:
            self._L.orc_destroy(self._h)
        except Exception:
            pass

    @property
    def t(self):
        return self._L.orc_t(self._h)

    @property
    def flow_t(self):
        return self._L.orc_flow_t(self._h)

    def set_state(self, u, v, p):
        a = [np.ascontiguousarray(x, dtype=np.float32).reshape(self.n, self.m) for x in (u, v, p)]
        self._L.orc_set_state(self._h, *a)

    def get_state(self):
        u = np.empty((self.n, self.m), np.float32)
        v = np.empty_like(u)
        p = np.empty_like(u)
        self._L.orc_get_state(self._h, u, v, p)
        return u, v, p

    def resume(self, path):
        rc = self._L.orc_resume(self._h, path.encode())
        if rc != 0:
            raise IOError("orc_resume(%s) -> %d" % (path, rc))

    def write(self, path):
        rc = self._L.orc_write(self._h, path.encode())
        if rc != 0:
            raise IOError("orc_write(%s) -> %d" % (path, rc))

    def update2(self, xi1_m=0.0, xi2_m=0.0):
        force = np.zeros(2, np.float32)
        it = (C.c_int * 2)()
        res = np.zeros(2, np.float32)
        self._L.orc_update2(self._h, xi1_m, xi2_m, force, it, res)
        return force, (it[0], it[1]), res

    def coeff(self, which):
        if isinstance(which, str):
            which = COEFF_NAMES.index(which)
        out = np.empty((self.n, self.m), np.float32)
        self._L.orc_get_coeff(self._h, which, out)
        return out

    def refresh_coeffs(self, dphi1, dphi2):
        self._L.orc_refresh_coeffs(self._h, dphi1, dphi2)

    def weights(self, x, y):
        w = np.zeros(3, np.float32)
        self._L.orc_weights(self._h, x, y, w)
        return w

    def cfl(self):
        return float(self._L.orc_cfl(self._h))

    def vorticity(self):
        out = np.empty((self.n, self.m), np.float32)
        self._L.orc_vorticity(self._h, out)
        return out

    def probes(self, num=32):
        out = np.empty(num, np.float32)
        self._L.orc_probes(self._h, num, out)
        return out

    def force(self):
        f = np.zeros(2, np.float32)
        self._L.orc_force(self._h, f)
        return f
