"""Are the workgroups of one CU in phase?  Prints start-of-step-8 and end times of envs b, b+256, b+512, b+768."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlfluidcontrol_amd import build, env  # noqa: E402

build.build()
stag = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
E = 1024
res = int(os.environ.get("RES", "16"))  # RES=32 RE=400: the 512x256 grid
g = env.AFCEnv(num_envs=E, resolution=res, reynolds=int(os.environ.get("RE", "100")), stagger_us=stag, dbg=8)
act = np.zeros((E, 2), np.float32)
g.agent_step(act)
obs = g.agent_step(act)[0].astype(np.float64) / 100.0  # us
t0 = obs[:, 0].min()
for b in (0, 1, 8, 100):
    print("stagger %#x envs %d+256k: step8 start %s  end %s" % (stag, b, np.round((obs[b::256, 0] - t0) % 167772.16, 0), np.round((obs[b::256, 1] - t0) % 167772.16, 0)))
st = (obs[:, 0] - t0) % 167772.16
en = (obs[:, 1] - t0) % 167772.16
print("  step-8 start spread: min %.0f max %.0f std %.0f us; end: min %.0f max %.0f" % (st.min(), st.max(), st.std(), en.min(), en.max()))
s4, e4 = st.reshape(4, 256), en.reshape(4, 256)  # [mate k][CU slot b]: env b + 256 k
print("  mean over the 256 CU slots, by mate (dispatch order): step-8 start %s   end %s   (launch ends at %.0f)"
      % (np.round(s4.mean(1), 0), np.round(e4.mean(1), 0), en.max()))
print("  workgroup slots idle before the launch ends: %.0f us on average (the second half of the launch, steps 8-15, takes %.0f us)"
      % ((en.max() - en).mean(), en.max() - st.min()))
print("  end of the launch by XCD (env %% 8), mean over its 128 workgroups: %s" % np.round([en[x::8].mean() for x in range(8)], 0))
cu_end = e4.max(0)  # when a CU slot's last mate ends
print("  last mate of a CU slot ends: min %.0f  median %.0f  max %.0f" % (cu_end.min(), np.median(cu_end), cu_end.max()))
