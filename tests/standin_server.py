"""TF-free stand-in for the reference's agent process: exposes the 8 XML-RPC methods of
server/server.py:51-59 with the same argument/return types, a deterministic "policy", the
reference's reward bookkeeping (server.py:61-65,148-165) and a full call log.  Test double only."""
import threading
from xmlrpc.server import SimpleXMLRPCServer

import numpy as np


class StandInServer:
    def __init__(self, port=0):
        self.server = SimpleXMLRPCServer(("127.0.0.1", port), logRequests=False, allow_none=True)
        self.port = self.server.server_address[1]
        self.calls = []
        self.state_record, self.action_record, self.rewards = [], [], []
        for name in ("init", "start_episode", "request_stochastic_action", "request_deterministic_action", "train",
                     "save", "save_eval", "restore"):
            self.server.register_function(getattr(self, "_" + name), name)
        self.thread = threading.Thread(target=self.server.serve_forever, daemon=True)

    def start(self):
        self.thread.start()
        return self

    def stop(self):
        self.server.shutdown()
        self.server.server_close()

    def _init(self, episode_count):
        self.calls.append(("init", episode_count))
        return False  # "freshly reset" (server.py:92-100)

    def _start_episode(self, raw):
        self.calls.append(("start_episode", raw))
        self.state_record, self.action_record = [], []
        return True

    def _request(self, raw, stochastic):
        self.calls.append(("request_stochastic_action" if stochastic else "request_deterministic_action", raw))
        state = np.array([float(x) for x in raw.split("_")], dtype=float)[None, :]  # server.py:118
        k = len(self.action_record)
        action = np.array([[np.sin(0.7 * k + state[0, 0]), np.cos(1.3 * k) * 0.5]])
        action = np.clip(action, -1, 1)
        self.state_record.append(state)
        self.action_record.append(action)
        return "_".join(str(i) for i in action[0, :])  # server.py:137

    def _request_stochastic_action(self, raw):
        return self._request(raw, True)

    def _request_deterministic_action(self, raw):
        return self._request(raw, False)

    def _train(self, steps):
        self.calls.append(("train", steps))
        self.rewards = []
        for i in range(len(self.state_record) - 1):  # server.py:157-165
            a, s2 = self.action_record[i], self.state_record[i + 1]
            self.rewards.append(-s2[0, 1] - np.pi * (1 / 8) * 0.0097 * (3.66 ** 3) * np.sum(np.abs(a) ** 3))
        return True

    def _save(self, dummy=None):
        self.calls.append(("save", dummy))
        return True

    def _save_eval(self, episode_count, rep_count):
        self.calls.append(("save_eval", episode_count, rep_count))
        return True

    def _restore(self, episode_count):
        self.calls.append(("restore", episode_count))
        return True
