"""CPU restatement of the reference's episode driver: the body of draw() and setUpNewSim()
(clientLilypad/clientCFD.pde:35-56,93-114) around the C oracle.  TEST INFRASTRUCTURE ONLY.

Float32 throughout, like the Java sketch: `Cd = Cd/callLearn*2/resolution` etc.
"""
import numpy as np

from .oracle_py import Oracle

f32 = np.float32


class OracleDrawLoop:
    """Same interface as AFCEnv for one env: reset() / agent_step(action[1,2])."""

    num_envs = 1

    def __init__(self, resolution=24, Re=500, init_state=None, init_time=1, end_time=50, substeps=16,
                 act_scale=5.0, carry_quirk=False, reward_penalty=0.18675613444715936):
        self.res, self.Re = resolution, Re
        self.init_state = init_state
        self.initTime, self.Time = f32(init_time), f32(end_time)
        self.substeps, self.act_scale = substeps, f32(act_scale)
        self.carry_quirk = carry_quirk
        self.penalty = f32(reward_penalty)
        self.callLearn = substeps  # sketch globals, clientCFD.pde:12-13
        self.Cd = f32(0)
        self.Cl = f32(0)
        self.obs = np.zeros(2, np.float32)
        self.win = False
        self.reset()

    def reset(self, mask=None):  # setUpNewSim, clientCFD.pde:93-114
        self.o = Oracle(resolution=self.res, Re=self.Re)
        if self.init_state is not None:
            self.o.set_state(*self.init_state)
        self.t = f32(0)
        self.dt = f32(self.o.dt)
        if not self.carry_quirk:  # the reference leaks these across episodes (globals)
            self.callLearn = self.substeps
            self.Cd = f32(0)
            self.Cl = f32(0)

    def agent_step(self, action):
        a = np.asarray(action, np.float32).reshape(-1, 2)[0]
        xi1_m, xi2_m = self.act_scale * a[0], self.act_scale * a[1]  # clientCFD.pde:53-54
        steps = 0
        self.win = False
        while True:
            if not (self.t < self.Time):  # clientCFD.pde:36
                break
            force, _, _ = self.o.update2(float(xi1_m), float(xi2_m))
            steps += 1
            self.t = f32(self.t + f32(self.dt / f32(self.res)))  # AFCCylinder.pde:56
            if self.t > self.initTime:  # clientCFD.pde:39-47
                self.callLearn -= 1
                self.Cd = f32(self.Cd + force[0])
                self.Cl = f32(self.Cl + force[1])
                if self.callLearn <= 0:
                    self.callLearn = self.substeps
                    self.Cd = f32(f32(f32(self.Cd / f32(self.callLearn)) * f32(2)) / f32(self.res))
                    self.Cl = f32(f32(f32(self.Cl / f32(self.callLearn)) * f32(2)) / f32(self.res))
                    self.obs = np.array([self.Cl, self.Cd], np.float32)
                    if not self.carry_quirk:
                        self.Cd = f32(0)
                        self.Cl = f32(0)
                    self.win = True
                    break
        aa = np.abs(a)
        rew = f32(-self.obs[1] - self.penalty * f32(aa[0] * aa[0] * aa[0] + aa[1] * aa[1] * aa[1]))
        done = not (self.t < self.Time)
        return (self.obs[None, :].copy(), np.array([rew], np.float32), np.array([done], np.uint8),
                np.array([steps], np.int32))

    def window_complete(self):
        """1 if the last agent_step ended with a completed averaging window (callAction was reached)"""
        return np.array([1 if self.win else 0], np.uint8)
