"""Generates a developed-wake initial state for a BASELINE grid (cold start
u=(1,0), zero action, T=100 => 13 334 sim-steps) on the GPU and saves it as .npz + .bdim under
gpurun_out/.  Run on the GPU box; copy the .npz into tests/golden/ to commit it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Re = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 13334
g = env.AFCEnv(num_envs=1, resolution=res, reynolds=Re, end_time=1e9, init_time=-1.0)
done = 0
log = []
while done < nsteps:
    obs, rew, dn, ns = g.agent_step(np.zeros((1, 2), np.float32))
    done += int(ns[0])
    if int(ns[0]) == 0:
        raise SystemExit('agent_step made no progress')
    log.append((done, float(obs[0, 0]), float(obs[0, 1])))
    if len(log) % 100 == 0:
        print(log[-1], flush=True)
u, v, p = g.get_state()
os.makedirs("gpurun_out", exist_ok=True)
name = "gpurun_out/wake_res%d_re%d" % (res, Re)
np.savez_compressed(name + ".npz", u=u, v=v, p=p, steps=done, cl_cd=np.array(log, np.float32))
g.write(name + ".bdim")
print("saved", name, "steps", done, "last Cl,Cd", log[-1], "diverged", g.diverged())
