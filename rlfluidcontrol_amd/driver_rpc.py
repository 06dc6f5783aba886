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
"""
import xmlrpc.client

import numpy as np

from .javafmt import java_float_to_string, java_parse_float


class LilypadRPCDriver:
    def __init__(self, env, url="http://localhost:8000", train_steps=1000, deterministic=False, log=None):
        self.env = env
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
            if done[0] and ns[0] < 1:
                break
            if done[0]:  # the window that was running when t reached Time never completes
                break
            xi = self.call_action(obs[0, 0], obs[0, 1])
            action[:] = xi[None, :]
            n += 1
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
