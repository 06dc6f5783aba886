"""On-disk artefacts of the reference's sketch, written from the batched
env's outputs:

  SaveScalarLog   the per-sim-step text log `saved/<run>.txt` of `SaveScalar.addData03`
                  (SaveScalar.pde:28-41,61-72): t, force, xi1, xi2 and 32 surface-pressure probes,
                  numbers printed as Java's float->String does (javafmt.py)
  vorticity_png   the picture `test.display()` saves (`FloodPlot.display(flow.u.curl())`,
                  FloodPlot.pde:54-88 with range (-1,1), hue 240 -> 0, light mode) as a PNG

The .bdim checkpoint is handled by the library itself (`afc_load_state_bdim` / `afc_save_state_bdim`)
and the episode `.npz` by `learner_server.py`.
"""
import struct
import zlib

import numpy as np

from .javafmt import java_float_to_string as jf


class SaveScalarLog:
    HEADER = ("%% Force and pressure coefficients using processing viscous simulation\n\n"
              "%% Fellowing: t, force.x, force.y\n\n")

    def __init__(self, path, num_theta=32):
        self.f = open(path, "w")
        self.f.write(self.HEADER)  # SaveScalar.pde:28-33
        self.num_theta = num_theta

    def add_data03(self, t, r1, r2, force, probes):
        """SaveScalar.pde:61-72: `t fx fy r1 r2 p0 ... p31 ` (every number followed by one blank)"""
        probes = np.asarray(probes, np.float32).ravel()
        assert probes.size == self.num_theta
        head = " ".join(jf(x) for x in (t, force[0], force[1], r1, r2))
        self.f.write(head + " " + "".join(jf(p) + " " for p in probes) + "\n")

    def finish(self):
        self.f.flush()
        self.f.close()


def record_run(env, log, steps, xi=(0.0, 0.0), env_idx=0, act_scale=5.0):
    """What draw() logs each frame (clientCFD.pde:37-38): one sim-step of every env, then a line for env
    `env_idx` with test.t, test.xi1, test.xi2, test.force and the probes of test.flow.p."""
    xi = np.asarray(xi, np.float32)
    xi_m = np.tile(act_scale * xi, (env.num_envs, 1)).astype(np.float32)
    for _ in range(steps):
        force = env.update2(xi_m)
        log.add_data03(env.t[env_idx], xi[0], xi[1], force[env_idx], env.probes(log.num_theta)[env_idx])


def flood_colors(f, low=-1.0, high=1.0):
    """FloodPlot.colorScale (FloodPlot.pde:54-68), default mode: below mid-range hue 240 (blue) with
    saturation 2(0.5-i), above it hue 0 (red) with saturation 2(i-0.5), brightness 1."""
    i = np.clip((np.asarray(f, np.float64) - low) / (high - low), 0.0, 1.0)  # Scale.in
    sat = np.abs(i - 0.5) * 2.0
    rgb = np.empty(i.shape + (3,), np.float64)
    lo = i < 0.5
    # HSB with B = 1: blue (240 deg) -> (1-s, 1-s, 1); red (0 deg) -> (1, 1-s, 1-s)
    rgb[..., 0] = np.where(lo, 1.0 - sat, 1.0)
    rgb[..., 1] = 1.0 - sat
    rgb[..., 2] = np.where(lo, 1.0, 1.0 - sat)
    return np.rint(rgb * 255.0).astype(np.uint8)


def write_png(path, rgb):
    """minimal PNG writer (8-bit RGB, zlib from the standard library)"""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def vorticity_png(path, vort, zoom=3, low=-1.0, high=1.0):
    """vort [n, m] as `afc_get_field(AFC_FIELD_VORT)` returns it (i along x, j along y); the sketch
    draws x to the right and y downwards, `zoom` screen pixels per cell (clientCFD.pde:10)."""
    img = flood_colors(np.asarray(vort, np.float32).T, low, high)  # rows = j (y), columns = i (x)
    img = np.repeat(np.repeat(img, zoom, axis=0), zoom, axis=1)
    write_png(path, np.ascontiguousarray(img))
    return img.shape[:2]
