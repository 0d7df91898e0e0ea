"""VecWRSN -- B independent WRSN environments stepped by the gfx950 kernels (one wavefront per environment).

Batched counterpart of the reference's `rl_env.WRSN.WRSN` (rl_env/WRSN.py:21): the same asynchronous
multi-agent protocol -- `reset()` / `step(agent_ids, actions)` return, per environment, the id of the mobile
charger that needs an action, its reward, a 4 x G x G observation, the terminal flag and the simulated time
-- with every array living in HBM as a torch tensor.  torch is used for device memory and streams only; all
environment arithmetic happens in csrc/ (HIP) behind the C-ABI of include/wrsn_hip.h.
"""
import numpy as np

from . import _lib
from .scenario import DEFAULT_MC_SPEC, MC_SPEC_KEYS, Scenario, load_mc_yaml


def _mc_spec_dict(agent_type):
    if agent_type is None:
        return dict(DEFAULT_MC_SPEC)
    if isinstance(agent_type, str):
        return load_mc_yaml(agent_type)
    return {k: float(agent_type[k]) for k in MC_SPEC_KEYS}


class VecWRSN:
    """Batched WRSN environment on one MI355X.

    scenarios : list of `Scenario` (one per environment; the same object may repeat)
    agent_type: dict / YAML path of the charger parameters (mc_types/default.yaml), None = shipped defaults
    num_agent : number of mobile chargers per environment (`num_agent`, WRSN.py:26)
    auto_reset: an environment whose last return was terminal is reset by the next `step` (status 3)
    reuse_obs : False (default): every returned row of `state` is rendered in full and the caller may do with the tensor what it likes.
                True: the caller promises to treat `state` as READ-ONLY (clone before normalising / clipping in place) -- map 1 of a row,
                the node map, which depends on node state only, is then not re-rendered by a step that returns at the instant it was
                called (`wrsn_set_obs_reuse`, ~40 % of the random-policy benchmark's returns); bit-identical results when the promise is
                kept, stale / corrupted map 1 when it is not.  bench.py, bench_ippo.py and the diagnostics opt in.
    step_budget: 0 = every `step` runs each WRSN.step to its end (the reference's blocking call).  > 0 bounds the work
                of one launch per environment (units of ~400 cycles counted per simulated second / service / exact second): an
                environment whose step is still in flight reports status 4 / agent_id -1 and simply goes on in the
                next `step` (its agent_id / action row is ignored).  Requests are identical either way; only the
                launch they appear in differs, so a batch no longer waits for its slowest environment.
    step_deadline_us: > 0 = time-sliced launches (`wrsn_set_step_deadline`): every launch lasts about that long; the waves walk the
                environments in a cyclic order, an environment whose step is not finished (or not even begun: its action then waits in a
                latch inside the library) reports status 4 / agent_id -1 and goes on in the following launches.  Same requests; which
                launch reports one depends on timing.  May be combined with a step budget (a cap per visit).
    """

    def __init__(self, scenarios, agent_type=None, num_agent=3, map_size=100, warm_up_time=100, device="cuda:0",
                 auto_reset=False, render=True, max_degree=0, max_cover=0, step_budget=0, reuse_obs=False, step_deadline_us=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("VecWRSN needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("VecWRSN runs on HIP devices only")
        scenarios = list(scenarios)
        if not scenarios or not all(isinstance(s, Scenario) for s in scenarios):
            raise TypeError("scenarios must be a non-empty list of Scenario")
        self.scenarios = scenarios
        self.mc_spec = _mc_spec_dict(agent_type)
        self.num_env = len(scenarios)
        self.num_agent = int(num_agent)
        self.map_size = int(map_size)
        self.warm_up_time = float(warm_up_time)
        self.auto_reset = bool(auto_reset)
        self.render = bool(render)
        self.n_node = max(s.n_node for s in scenarios)
        self.n_target = max(s.n_target for s in scenarios)
        lib = _lib.load()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev_index):
            self._h = _lib.RawHandle(lib, self.num_env, self.n_node, self.n_target, self.num_agent, self.map_size,
                                     self.warm_up_time, dev_index, max_degree, max_cover)
            self._h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
            if reuse_obs and self.render:
                self._h.set_obs_reuse(True)
            self.step_budget = int(step_budget)
            if self.step_budget:
                self._h.set_step_budget(self.step_budget)
            self.step_deadline_us = int(step_deadline_us)
            if self.step_deadline_us:
                self._h.set_step_deadline(self.step_deadline_us)
            # topology build + warm-up + snapshot happen on the device inside set_scenario
            self._h.set_scenarios(scenarios, self.mc_spec)
            B, G = self.num_env, self.map_size
            self.agent_id = torch.full((B,), -1, dtype=torch.int32, device=self.device)
            self.reward = torch.zeros(B, dtype=torch.float64, device=self.device)
            self.terminal = torch.zeros(B, dtype=torch.uint8, device=self.device)
            self.now = torch.zeros(B, dtype=torch.float64, device=self.device)
            self.status = torch.zeros(B, dtype=torch.int32, device=self.device)
            self.state = torch.zeros((B, 4, G, G), dtype=torch.float32, device=self.device) if self.render else None
            self._in_agent = torch.zeros(B, dtype=torch.int32, device=self.device)
            self._in_action = torch.zeros((B, 3), dtype=torch.float64, device=self.device)

    # -- plumbing -----------------------------------------------------------------------------------------
    def _out_ptrs(self):
        return dict(agent_id=self.agent_id.data_ptr(), reward=self.reward.data_ptr(), terminal=self.terminal.data_ptr(),
                    now=self.now.data_ptr(), status=self.status.data_ptr(),
                    obs=(self.state.data_ptr() if self.state is not None else 0))

    def _result(self):
        return {"agent_id": self.agent_id, "reward": self.reward, "terminal": self.terminal, "now": self.now,
                "status": self.status, "state": self.state}

    def _bind_stream(self):
        self._h.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    # -- API ----------------------------------------------------------------------------------------------
    def reset(self, mask=None):
        """WRSN.reset (WRSN.py:41-83) for all environments, or those with mask != 0.  Returns the request
        tensors (views on internal buffers: agent_id, reward, terminal, now, status, state)."""
        self._bind_stream()
        mptr = 0
        if mask is not None:
            self._mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            mptr = self._mask.data_ptr()
        self._h.reset(mptr, **self._out_ptrs())
        return self._result()

    def step(self, agent_ids, actions):
        """WRSN.step (WRSN.py:289-330) for every environment.

        agent_ids: int tensor [B]; >= 0 gives `actions[b]` to that charger, -1 means "just run" (agent_id=None in
                   the reference), -2 leaves the environment untouched.
        actions  : float tensor [B,3] in [0,1] (clipped inside, WRSN.py:299); density_map=False path."""
        t = self.torch
        self._bind_stream()
        # Arrays that already are what the C-ABI takes (int32 / float64, contiguous, on this device) go in as they are -- including this
        # object's own `agent_id` output tensor: a launch reads row e of the inputs and writes row e of the outputs from the one block
        # that owns environment e.  Anything else is converted into the staging buffers (two small copy kernels per call).
        a = agent_ids
        if not (a.dtype == t.int32 and a.device == self.device and a.is_contiguous() and a.numel() == self.num_env):
            self._in_agent.copy_(a.to(device=self.device, dtype=t.int32).reshape(-1)); a = self._in_agent
        x = actions
        if not (x.dtype == t.float64 and x.device == self.device and x.is_contiguous() and x.numel() == 3 * self.num_env):
            self._in_action.copy_(x.to(device=self.device, dtype=t.float64).reshape(-1, 3)); x = self._in_action
        self._keep_in = (a, x)                                # alive until the launch has run
        self._h.step(a.data_ptr(), x.data_ptr(), self.auto_reset, **self._out_ptrs())
        return self._result()

    def render_state(self, agent_ids, out=None):
        """get_state(agent) (WRSN.py:130-186) for arbitrary agents; rows with agent < 0 are left untouched."""
        t = self.torch
        self._bind_stream()
        a = agent_ids.to(device=self.device, dtype=t.int32).contiguous()
        if out is None:
            out = t.zeros((self.num_env, 4, self.map_size, self.map_size), dtype=t.float32, device=self.device)
        self._h.render(a.data_ptr(), out.data_ptr())
        return out

    def set_step_budget(self, work_units):
        self.step_budget = int(work_units)
        self._h.set_step_budget(self.step_budget)

    def density_to_action(self, agent_ids, dmaps, out=None):
        """WRSN.density_map_to_action (WRSN.py:229-287, with the exp-normalisation of WRSN.py:293-296) on the device:
        dmaps [B, G, G] -> actions [B, 3] float64 for `step`.  Rows with agent < 0 are left untouched."""
        t = self.torch
        self._bind_stream()
        a = agent_ids.to(device=self.device, dtype=t.int32).contiguous()
        m = dmaps.to(device=self.device, dtype=t.float64).contiguous()
        if out is None:
            out = t.zeros((self.num_env, 3), dtype=t.float64, device=self.device)
        self._h.density_action(a.data_ptr(), m.data_ptr(), out.data_ptr())
        return out

    def rollout_table(self, zero_after=False, out=None):
        """[B, M + 3] float64 device tensor accumulated inside the step kernel: sum of rewards per charger, finished
        episodes, sum of lifetimes (env.now at terminal), completed WRSN.step calls -- the layout of
        `sharding.RolloutStats` (pass it to `RolloutStats.gather_table`)."""
        t = self.torch
        self._bind_stream()
        if out is None:
            out = t.empty((self.num_env, self.num_agent + 3), dtype=t.float64, device=self.device)
        self._h.rollout_table(out.data_ptr(), zero_after)
        return out

    def synchronize(self):
        self._h.sync()

    # -- read-only views for tests / logging (host copies) --------------------------------------------------
    def nodes(self):
        p = self._h.peek
        return {"energy": p(_lib.PEEK_NODE_ENERGY), "cs": p(_lib.PEEK_NODE_CS), "rr": p(_lib.PEEK_NODE_RR),
                "status": p(_lib.PEEK_NODE_STATUS), "level": p(_lib.PEEK_NODE_LEVEL)}

    def topology(self):
        p = self._h.peek
        return {"degree": p(_lib.PEEK_NODE_DEGREE), "n_cover": p(_lib.PEEK_NODE_NCOVER), "direct": p(_lib.PEEK_NODE_DIRECT)}

    def targets_active(self):
        """Network.targets_active (Network.py:9, 45-55) per environment: int32 [B, T]."""
        return self._h.peek(_lib.PEEK_TARGETS_ACTIVE)

    def mcs(self):
        a = self._h.peek(_lib.PEEK_MC)
        return {k: a[:, :, i].copy() for i, k in enumerate(_lib.MC_FIELDS) if not k.startswith("_")}

    def env_info(self):
        a = self._h.peek(_lib.PEEK_ENV)
        return {k: a[:, i].copy() for i, k in enumerate(_lib.ENV_FIELDS)}

    def counters(self):
        return self._h.counters()

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._h.close()
            self._h = None
