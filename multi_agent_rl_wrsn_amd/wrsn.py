"""`WRSN` -- single-environment facade with the reference's exact constructor and request-dict protocol
(rl_env/WRSN.py:21-39, 41-83, 289-330), so `runner/IPPO.py`, `runner/PPO.py` and `runner/checkRL.py` only change
their import.  It is a B = 1 `VecWRSN`: all dynamics run in the HIP kernels; this file is host glue.

What is mirrored: the constructor signature, `reset()` / `step(agent_id, input_action)` return dicts with keys
agent_id / prev_state / input_action / action / reward / state / terminal / info, float64 4 x G x G states,
`num_agent`, `env.now`, `net` / `agents` read-only views, the `observation_space` / `action_space` boxes, and the
re-seeding of the global `random` / `numpy.random` generators on every reset (NetworkIO.py:22-24).
Deliberate deviations: when every charger is dead the reference never returns -- here `step` returns a
terminal request; with `density_map=True` the charging spot inside the arg-max box comes from a deterministic search on
the device instead of SciPy's L-BFGS-B (`wrsn_density_action`).
"""
import random

import numpy as np

from . import _lib
from .scenario import load_mc_yaml, load_scenario_yaml
from .vec_env import VecWRSN


class Box:
    """The slice of gym.spaces.Box consumers touch (WRSN.py:31-32, 299)."""

    def __init__(self, low, high, shape, dtype=np.float64):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class _Clock:
    def __init__(self, owner):
        self._o = owner

    @property
    def now(self):
        return self._o._now


class _NodeView:
    def __init__(self, i, xy, spec):
        self.id = i
        self.location = np.array(xy, dtype=np.float64)
        self.capacity = float(spec["capacity"]); self.threshold = float(spec["threshold"])
        self.energy = self.capacity; self.energyCS = 0.0; self.energyRR = 0.0; self.status = 1; self.level = None


class _TargetView:                                           # Target.py:1-4
    def __init__(self, i, xy):
        self.id = i
        self.location = [float(xy[0]), float(xy[1])]


class _BaseStationView:
    def __init__(self, xy):
        self.location = np.array(xy, dtype=np.float64)


class _NetView:
    """Read-only picture of `Network` (Network.py:3-92) refreshed from the device on access."""

    def __init__(self, owner):
        self._o = owner
        sc = owner.scenario
        self.env = owner.env
        self.baseStation = _BaseStationView(sc.bs_xy)
        self.listNodes = [_NodeView(i, sc.node_xy[i], sc.node_spec) for i in range(sc.n_node)]
        self.listTargets = [_TargetView(i, sc.target_xy[i]) for i in range(sc.n_target)]
        self.max_time = sc.max_time
        self.frame = sc.frame()
        self.nodes_density = sc.n_node / ((self.frame[1] - self.frame[0]) * (self.frame[3] - self.frame[2]))

    def refresh(self):
        nd = self._o.vec.nodes()
        for i, n in enumerate(self.listNodes):
            n.energy = float(nd["energy"][0, i]); n.energyCS = float(nd["cs"][0, i]); n.energyRR = float(nd["rr"][0, i])
            n.status = int(nd["status"][0, i]); n.level = int(nd["level"][0, i])
        return self

    @property
    def alive(self):
        return int(self._o.vec.env_info()["alive"][0])

    @property
    def targets_active(self):                                # Network.py:9; read by runner/checkRL.py:25-27
        return [int(v) for v in self._o.vec.targets_active()[0, :self._o.scenario.n_target]]

    def check_targets(self):                                 # Network.py:84-85
        return min(self.targets_active)

    def check_nodes(self):                                   # Network.py:87-92
        return int((self._o.vec.nodes()["status"][0] == 0).sum())


class _AgentView:
    def __init__(self, i, spec):
        self.id = i
        self.capacity = spec["capacity"]; self.threshold = spec["threshold"]; self.velocity = spec["velocity"]
        self.pm = spec["pm"]; self.chargingRange = spec["charging_range"]; self.alpha = spec["alpha"]; self.beta = spec["beta"]
        self.epsilon = spec["epsilon"]
        self.location = None; self.energy = None; self.status = 1; self.cur_action_type = "moving"; self.cur_phy_action = None


class WRSN:
    def __init__(self, scenario_path, agent_type_path, num_agent, map_size=100, warm_up_time=100, density_map=False,
                 device="cuda:0"):
        self.scenario = load_scenario_yaml(scenario_path)
        self.agent_phy_para = load_mc_yaml(agent_type_path)
        self.num_agent = num_agent
        self.map_size = map_size
        self.density_map = density_map
        self.warm_up_time = warm_up_time
        self.epsilon = 1e-9
        self.observation_space = Box(0.0, 1.0, (4, map_size, map_size), np.float64)
        self.action_space = Box(0.0, 1.0, (3,), np.float64)
        self.agents_input_action = [None for _ in range(num_agent)]
        self.agents_action = [None for _ in range(num_agent)]
        self.agents_prev_state = [None for _ in range(num_agent)]
        self.vec = VecWRSN([self.scenario], self.agent_phy_para, num_agent, map_size, warm_up_time, device=device)
        self._now = 0.0
        self.env = _Clock(self)
        self.net = _NetView(self)
        self.agents = [_AgentView(i, self.agent_phy_para) for i in range(num_agent)]
        info = self.vec.env_info()
        self.moving_time_max = float(info["moving_time_max"][0])
        self.charging_time_max = float(info["charging_time_max"][0])
        self.avg_nodes_agent = float(info["avg_nodes_agent"][0])
        self.reset()

    # -- helpers ------------------------------------------------------------------------------------------
    def _refresh_agents(self):
        m = self.vec.mcs()
        for i, a in enumerate(self.agents):
            a.location = np.array([m["loc_x"][0, i], m["loc_y"][0, i]]); a.energy = float(m["energy"][0, i])
            a.status = int(m["status"][0, i]); a.cur_action_type = "charging" if m["type_charging"][0, i] > 0 else "moving"
            a.cur_phy_action = [float(m["cur_x"][0, i]), float(m["cur_y"][0, i]), float(m["cur_t"][0, i])]
        return self.agents

    def _info(self):
        return [self.net.refresh(), self._refresh_agents()]

    def down_mapping(self, location):                        # WRSN.py:86-88
        f = self.net.frame
        return np.array([(location[0] - f[0]) / (f[1] - f[0]), (location[1] - f[2]) / (f[3] - f[2])])

    def up_mapping(self, down_map):                          # WRSN.py:91-93
        f = self.net.frame
        return np.array([down_map[0] * (f[1] - f[0]) + f[0], down_map[1] * (f[3] - f[2]) + f[2]])

    def get_state(self, agent_id):                           # WRSN.py:130-186, rendered on the device
        t = self.vec.torch
        ids = t.tensor([int(agent_id)], dtype=t.int32)
        return self.vec.render_state(ids)[0].to("cpu").numpy().astype(np.float64)

    # -- API ----------------------------------------------------------------------------------------------
    def reset(self):
        np.random.seed(self.scenario.seed)                   # NetworkIO.py:22-24 side effect
        random.seed(self.scenario.seed)
        r = self.vec.reset()
        self.vec.synchronize()
        self._now = float(r["now"][0])
        terminal = bool(r["terminal"][0])
        aid = int(r["agent_id"][0])
        bs_action = np.reshape(np.append(self.down_mapping(self.scenario.bs_xy), 0), (3,))
        for i in range(self.num_agent):
            self.agents_action[i] = bs_action.copy()
        if aid < 0:
            return {"agent_id": None, "prev_state": None, "input_action": None, "action": None, "reward": None,
                    "state": None, "terminal": terminal, "info": self._info()}
        state0 = r["state"][0].to("cpu").numpy().astype(np.float64)
        for i in range(self.num_agent):
            self.agents_prev_state[i] = state0 if i == aid else self.get_state(i)
        return {"agent_id": aid, "prev_state": self.agents_prev_state[aid], "input_action": self.agents_input_action[aid],
                "action": self.agents_action[aid], "reward": 0.0, "state": self.agents_prev_state[aid],
                "terminal": terminal, "info": self._info()}

    def step(self, agent_id, input_action, _action3=None):
        """WRSN.step (WRSN.py:289-330).  `_action3` is a test hook, not part of the reference's signature: with `density_map=True`
        it replaces the 3-vector derived from the map (the charging spot inside the arg-max box is the one result of the path whose
        parity is unpinned), so that a fixture recorded from the reference can be followed decision by decision."""
        t = self.vec.torch
        act3 = np.zeros(3)
        if agent_id is not None:
            action = np.array(input_action)
            self.agents_input_action[agent_id] = action.copy()
            if self.density_map:                             # WRSN.py:293-297 (normalisation included) on the device
                action = self.density_map_to_action(action, agent_id) if _action3 is None else np.asarray(_action3, dtype=np.float64)
            action = np.clip(action, self.action_space.low, self.action_space.high)
            self.agents_action[agent_id] = action
            self.agents_prev_state[agent_id] = self.get_state(agent_id)     # WRSN.py:303
            act3 = np.asarray(action, dtype=np.float64).reshape(3)
        ids = t.tensor([-1 if agent_id is None else int(agent_id)], dtype=t.int32)
        r = self.vec.step(ids, t.tensor(act3.reshape(1, 3), dtype=t.float64))
        self.vec.synchronize()
        while int(r["status"][0]) == 4:                      # only with a step budget: the step is still in flight
            r = self.vec.step(ids, t.tensor(act3.reshape(1, 3), dtype=t.float64))
            self.vec.synchronize()
        self._now = float(r["now"][0])
        status = int(r["status"][0])
        if status < 0:
            raise RuntimeError("environment error status %d (connection-list capacity exceeded?)" % status)
        if bool(r["terminal"][0]):                           # WRSN.py:312-320 (and the all-chargers-dead deviation)
            return {"agent_id": None, "prev_state": None, "input_action": None, "action": None, "reward": None,
                    "state": None, "terminal": True, "info": self._info()}
        aid = int(r["agent_id"][0])
        if aid < 0:
            return None                                      # the reference falls off the end of step() (WRSN.py:321-330)
        state = r["state"][0].to("cpu").numpy().astype(np.float64)
        return {"agent_id": aid, "prev_state": self.agents_prev_state[aid], "input_action": self.agents_input_action[aid],
                "action": self.agents_action[aid], "reward": float(r["reward"][0]), "state": state, "terminal": False,
                "info": self._info()}

    # -- density-map action extraction (WRSN.py:229-287) on the device ------------------------------------------
    def density_map_to_action(self, dmap, id):
        t = self.vec.torch
        ids = t.tensor([int(id)], dtype=t.int32)
        dm = t.as_tensor(np.asarray(dmap, dtype=np.float64).reshape(1, self.map_size, self.map_size))
        return self.vec.density_to_action(ids, dm)[0].to("cpu").numpy()
