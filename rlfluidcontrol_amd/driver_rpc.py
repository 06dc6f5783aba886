"""The env side of the reference's agent/environment boundary: our counterpart of the
Processing sketch clientLilypad/clientCFD.pde, speaking the same XML-RPC calls to an UNCHANGED
server/server.py (or any server exposing the same 8 methods, server.py:51-59).

    setup()        clientCFD.pde:18-33   init(-1); setUpNewSim
    draw()         clientCFD.pde:35-85   step, average 16 forces, "Cl_Cd" -> "a1_a2", episode end
    setUpNewSim()  clientCFD.pde:93-114  new AFCCylinder from the reset state; start_episode(-1)
    callAction()   clientCFD.pde:117-133 request_stochastic_action; any failure -> action (0,0)

The simulation itself is the batched HIP environment (`AFCEnv.agent_step` runs the body of
draw() between two action requests on the GPU); this module only does the protocol.  The server
is stateful for exactly one episode, so the driver uses env index 0 of the batch.

Drop-in note: the sketch never clears its `Cd`/`Cl` globals after a request (clientCFD.pde:41-47), so every
window after the first starts from the previous normalised mean (~0.5 % carry) and the sums leak across
episodes.  An unchanged server.py sees exactly the reference client's strings only when the env is built with
`carry_quirk=1`: use `LilypadRPCDriver.reference_env(...)`.  `carry_quirk=0` (the library default, used by
train.py and bench.py) is the cleaned-up averaging and is a stated deviation from the reference client.
"""
import xmlrpc.client

import numpy as np

from .javafmt import java_float_to_string, java_parse_float


class LilypadRPCDriver:
    @staticmethod
    def reference_env(resolution=24, reynolds=500, **kw):
        """One HIP env configured like the reference sketch, including its Cd/Cl carry (clientCFD.pde:41-47)."""
        from .env import AFCEnv
        kw.setdefault("carry_quirk", 1)
        return AFCEnv(num_envs=1, resolution=resolution, reynolds=reynolds, **kw)

    def __init__(self, env, url="http://localhost:8000", train_steps=1000, deterministic=False, log=None):
        self.env = env
        cfg = getattr(env, "cfg", None)
        if cfg is not None and not getattr(cfg, "carry_quirk", 1) and log is not None:
            log("note: env built with carry_quirk=0 -- observations deviate from the reference client's by its "
                "Cd/Cl carry (use LilypadRPCDriver.reference_env for drop-in strings)")
        self.client = xmlrpc.client.ServerProxy(url, allow_none=True)
        self.train_steps = train_steps
        self.method = "request_deterministic_action" if deterministic else "request_stochastic_action"
        self.log = log if log is not None else (lambda *a: None)
        self.sim_num = 1
        self.transcript = []  # (method, argument, result) of every call, for tests and debugging

    # --- the reference wraps every RPC in try/catch and carries on (clientCFD.pde:27-29,79-81) ---
    def _call(self, method, arg):
        try:
            res = getattr(self.client, method)(arg)
        except Exception as ex:  # noqa: BLE001 - mirror of `catch (Exception ex) { println(ex); }`
            self.log("rpc %s failed: %r" % (method, ex))
            res = None
        self.transcript.append((method, arg, res))
        return res

    def setup(self):
        self._call("init", -1)
        self.set_up_new_sim()

    def set_up_new_sim(self):
        self.env.reset()
        self._call("start_episode", -1)

    def call_action(self, cl, cd):
        arg = java_float_to_string(cl) + "_" + java_float_to_string(cd)
        res = self._call(self.method, arg)
        xi = np.zeros(2, np.float32)  # float[] XI = new float[2]  -> (0,0) on any failure
        try:
            parts = res.split("_")
            xi[0] = java_parse_float(parts[0])
            xi[1] = java_parse_float(parts[1])
        except Exception as ex:  # noqa: BLE001
            self.log("bad action %r: %r" % (res, ex))
        return xi

    def run_episode(self, max_requests=None):
        """One pass of draw() from t=0 to t>=Time.  Returns the number of action requests."""
        action = np.zeros((self.env.num_envs, 2), np.float32)
        n = 0
        while True:
            obs, _rew, done, ns = self.env.agent_step(action)
            # a window that completes on the very step that takes t to Time is still sent (draw() asks inside
            # the step and only ends the episode on the next frame, clientCFD.pde:36-56,67); a window cut short
            # by t >= Time is not
            fresh = (not done[0]) or bool(self.env.window_complete()[0])
            if fresh:
                xi = self.call_action(obs[0, 0], obs[0, 1])
                action[:] = xi[None, :]
                n += 1
            if done[0]:
                break
            if max_requests is not None and n >= max_requests:
                break
        return n

    def finish_episode(self):
        self._call("train", self.train_steps)   # clientCFD.pde:73-76
        self._call("save", self.train_steps)    # clientCFD.pde:77
        self.sim_num += 1
        self.set_up_new_sim()

    def run(self, episodes=1):
        self.setup()
        for _ in range(episodes):
            self.run_episode()
            self.finish_episode()
