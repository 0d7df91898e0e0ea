"""TEST INFRASTRUCTURE -- a scripted stand-in for the environment side of a roll-out: seeded agent order, rewards and episode lengths,
"states" that are 4-float tags (episode, request number, asking agent, 0.5).  It answers with the keys the reference's roll_out reads
(controller/ippo/IPPO.py:137-165) -- `prev_state` / `input_action` are what the asking agent saw / did at its previous decision, as
rl_env/WRSN.py returns them -- plus `detailed_rewards`, a key the reference's roll_out logs but its WRSN.step never returns (the
reason that method cannot run against the real environment).  Used by oracle/refharness/gen_rollout_golden.py, which runs the
reference's own roll_out on it, and by tests/test_ippo.py, which runs the build's bookkeeping on the same script."""
import types

import numpy as np


class ScriptEnv:
    def __init__(self, seed, n_agent):
        self.rng = np.random.RandomState(seed); self.n = n_agent; self.k = -1
        self.env = types.SimpleNamespace(now=0.0)             # `self.env.env.now` (IPPO.py:181)

    def _req(self, agent, reward, terminal):
        self.count += 1
        st = np.array([self.k, self.count, agent, 0.5], dtype=np.float32)
        self.cur_state = st
        return dict(agent_id=agent, state=st, prev_state=self.last_state[agent], input_action=self.last_action[agent],
                    action=[0.1 * agent, 0.2, 0.3], reward=reward, terminal=terminal, detailed_rewards=[reward, 0.0, 0.0], now=float(self.count))

    def reset(self):
        self.k += 1; self.count = 0; self.len = int(self.rng.randint(6, 14))
        self.last_state = [None] * self.n; self.last_action = [None] * self.n; self.env.now = 0.0
        return self._req(0, 0.0, False)

    def step(self, agent, action):
        self.last_state[agent] = self.cur_state; self.last_action[agent] = action
        nxt = int(self.rng.randint(self.n)); rew = float(np.float32(self.rng.randn())); term = self.count >= self.len
        self.env.now = float(self.count)
        return self._req(nxt, rew, term)


def scripted_policy(n, agent_id):
    """decision number n of the roll-out: (action, log-probability), both tagged"""
    return np.array([n, agent_id, 0.25], dtype=np.float32), np.float32(-n - 0.5)
