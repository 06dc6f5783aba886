"""Regenerates the committed golden fixtures.  Run HERE (needs /root/reference); the GPU
box only sees the committed .npz/.json outputs.

  init_bdim_res24.npz  <- /root/reference/clientLilypad/saved/init/init.bdim
      The reference's only numerical artifact on the hot path (BDIM.write format,
      BDIM.pde:226-237): header t, dt, then n*m lines "u, v, p" (i outer, j inner, ghosts
      included).  Stored as float32 arrays [n=386, m=194]; text->float32 by strtof (correctly
      rounded, same as Java Float.parseFloat).  This is DATA held by the reference, not source.
  server_utils_golden.json <- outputs of the importable reference module server/utils.py
      (KalmanFilter, ProcessNoise, ReplayBuffer) and of Server._reward_func (pure numpy),
      imported from /root/reference with a stub `tensorflow` module.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def bdim_to_npz():
    path = os.path.join(REF, "clientLilypad/saved/init/init.bdim")
    with open(path) as f:
        lines = f.read().split("\n")
    t = np.float32(lines[0])
    dt = np.float32(lines[1])
    n, m = 386, 194
    body = lines[2:2 + n * m]
    arr = np.array([[np.float32(x) for x in ln.split(",")] for ln in body], dtype=np.float32)
    u = arr[:, 0].reshape(n, m)
    v = arr[:, 1].reshape(n, m)
    p = arr[:, 2].reshape(n, m)
    # keep the first/last few raw text lines so the writer's formatting can be tested
    np.savez_compressed(os.path.join(HERE, "init_bdim_res24.npz"), t=t, dt=dt, u=u, v=v, p=p,
                        head_text=np.array(lines[:12]), tail_text=np.array(lines[2 + n * m - 4:2 + n * m]))
    print("init.bdim ->", n, m, t, dt)


def server_utils():
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "server"))
    import utils as ru  # reference module, numpy only
    out = {}
    kf = ru.KalmanFilter(Q=0.25, R=11) if "Q" in ru.KalmanFilter.__init__.__code__.co_varnames else ru.KalmanFilter()
    est = []
    for _ in range(6):
        est.append(np.asarray(kf.estimate(np.array([[1.0, 2.0]]))).ravel().tolist())
    out["kalman_const_1_2"] = est
    np.random.seed(0)
    pn = ru.ProcessNoise(2)
    out["process_noise_seed0"] = [np.asarray(pn.next()).ravel().tolist() for _ in range(4)]
    rb = ru.ReplayBuffer(2, 2, 4)
    for k in range(6):
        rb.store(np.array([k, k + 0.5]), np.array([0.1 * k, -0.1 * k]), float(k), np.array([k + 1, k + 1.5]), 0)
    out["replay_after_6"] = {"ptr": int(rb.ptr), "size": int(rb.size),
                             "obs1_col0": rb.obs1_buf[:, 0].tolist() if hasattr(rb, "obs1_buf") else None}
    # reward (server.py:61-65) needs a stub tensorflow to import
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    try:
        import server as rs
        cases = []
        for s, a, s2 in [((0.1, 1.1), (0.5, -0.5), (0.05, 1.2)), ((0, 1), (0, 0), (0, 0.9)), ((0.2, 1.3), (1, 1), (0.1, 1.0))]:
            r = rs.Server._reward_func(None, np.array([s]), np.array([a]), np.array([s2]))
            cases.append({"s": s, "a": a, "s2": s2, "r": float(np.asarray(r).ravel()[0])})
        out["reward_cases"] = cases
    except Exception as e:  # keep going: the other vectors are still valid
        out["reward_cases_error"] = repr(e)
    with open(os.path.join(HERE, "server_utils_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("server utils golden written:", list(out))


if __name__ == "__main__":
    bdim_to_npz()
    server_utils()
