"""Are the workgroups of one CU in phase?  Prints start-of-step-8 and end times of envs b, b+256, b+512, b+768."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
stag = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
E = 1024
g = env.AFCEnv(num_envs=E, resolution=16, reynolds=100, stagger_us=stag, dbg=8)
act = np.zeros((E, 2), np.float32)
g.agent_step(act)
obs = g.agent_step(act)[0].astype(np.float64) / 100.0  # us
t0 = obs[:, 0].min()
for b in (0, 1, 8, 100):
    print("stagger %#x envs %d+256k: step8 start %s  end %s" % (stag, b, np.round((obs[b::256, 0] - t0) % 167772.16, 0), np.round((obs[b::256, 1] - t0) % 167772.16, 0)))
st = (obs[:, 0] - t0) % 167772.16
en = (obs[:, 1] - t0) % 167772.16
print("  step-8 start spread: min %.0f max %.0f std %.0f us; end: min %.0f max %.0f" % (st.min(), st.max(), st.std(), en.min(), en.max()))
