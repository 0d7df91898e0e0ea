"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/libwrsn_oracle.so (the C restatement
of the reference's WRSN step path, see wrsn_oracle.c).  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg only; the product package
(`multi_agent_rl_wrsn_amd`) never imports it."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwrsn_oracle.so")
_lib = None


class WoCfg(C.Structure):
    _fields_ = [("n_node", C.c_int32), ("n_target", C.c_int32), ("n_mc", C.c_int32), ("map_size", C.c_int32),
                ("warm_up_time", C.c_double)] + \
               [(k, C.c_double) for k in ("capacity", "threshold", "com_range", "sen_range", "prob_gp",
                                          "package_size", "er", "et", "efs", "emp", "max_time",
                                          "mc_capacity", "mc_threshold", "velocity", "pm", "charging_range",
                                          "alpha", "beta", "epsilon")]


class WoOut(C.Structure):
    _fields_ = [("agent_id", C.c_int32), ("terminal", C.c_int32), ("status", C.c_int32), ("pad", C.c_int32),
                ("reward", C.c_double), ("now", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "wrsn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libwrsn_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.wo_create.restype = C.c_void_p
        L.wo_create.argtypes = [C.POINTER(WoCfg), dp, dp, dp]
        L.wo_destroy.argtypes = [C.c_void_p]
        L.wo_reset.argtypes = [C.c_void_p, C.POINTER(WoOut)]
        L.wo_step.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(WoOut)]
        L.wo_get_state.argtypes = [C.c_void_p, C.c_int, dp]
        L.wo_peek_nodes.argtypes = [C.c_void_p, dp, dp, dp, ip, ip]
        L.wo_peek_mcs.argtypes = [C.c_void_p, dp]
        L.wo_peek_env.argtypes = [C.c_void_p, dp]
        L.wo_peek_topology.argtypes = [C.c_void_p, ip, ip, ip]
        L.wo_peek_topology.restype = C.c_int
        L.wo_peek_targets.argtypes = [C.c_void_p, ip]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


MC_FIELDS = ("loc_x", "loc_y", "energy", "status", "type_charging", "cur_x", "cur_y", "cur_t", "n_conn",
             "excl", "prev_minfit", "act0", "act1", "act2")


class OracleWRSN:
    """One environment of the float64 CPU oracle.

    node_xy [N,2], target_xy [T,2], bs_xy [2]; node_spec / mc_spec are the reference's YAML
    dictionaries (`node_phy_spe`, mc_types/default.yaml).  step()/reset() return a dict with the
    reference's request keys (WRSN.py:68-83, 323-330) restricted to numeric content.
    """

    def __init__(self, node_xy, target_xy, bs_xy, node_spec, mc_spec, max_time, num_agent,
                 map_size=100, warm_up_time=100):
        self.node_xy = np.ascontiguousarray(node_xy, dtype=np.float64).reshape(-1, 2)
        self.target_xy = np.ascontiguousarray(target_xy, dtype=np.float64).reshape(-1, 2)
        self.bs_xy = np.ascontiguousarray(bs_xy, dtype=np.float64).reshape(2)
        self.N, self.T, self.M, self.G = len(self.node_xy), len(self.target_xy), int(num_agent), int(map_size)
        cfg = WoCfg()
        cfg.n_node, cfg.n_target, cfg.n_mc, cfg.map_size = self.N, self.T, self.M, self.G
        cfg.warm_up_time = float(warm_up_time)
        for k in ("capacity", "threshold", "com_range", "sen_range", "prob_gp", "package_size", "er", "et", "efs", "emp"):
            setattr(cfg, k, float(node_spec[k]))
        if float(node_spec["prob_gp"]) != 1.0:
            raise ValueError("oracle supports prob_gp == 1 only (every shipped scenario; see DESIGN.md)")
        cfg.max_time = float(max_time)
        cfg.mc_capacity, cfg.mc_threshold = float(mc_spec["capacity"]), float(mc_spec["threshold"])
        cfg.velocity, cfg.pm, cfg.charging_range = float(mc_spec["velocity"]), float(mc_spec["pm"]), float(mc_spec["charging_range"])
        cfg.alpha, cfg.beta, cfg.epsilon = float(mc_spec["alpha"]), float(mc_spec["beta"]), float(mc_spec["epsilon"])
        self._cfg = cfg
        self._h = lib().wo_create(C.byref(cfg), _dp(self.node_xy), _dp(self.target_xy), _dp(self.bs_xy))
        if not self._h:
            raise RuntimeError("wo_create failed")
        self.last = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().wo_destroy(self._h)
            self._h = None

    def _pack(self, out, with_state=True):
        d = {"agent_id": (None if out.agent_id < 0 else int(out.agent_id)), "reward": float(out.reward),
             "terminal": bool(out.terminal), "now": float(out.now), "status": int(out.status), "state": None}
        if with_state and d["agent_id"] is not None and not d["terminal"]:
            d["state"] = self.get_state(d["agent_id"])
        self.last = d
        return d

    def reset(self, with_state=True):
        out = WoOut()
        lib().wo_reset(self._h, C.byref(out))
        d = self._pack(out, with_state=False)
        if with_state and d["agent_id"] is not None:
            d["state"] = self.get_state(d["agent_id"])
        return d

    def step(self, agent_id, action, with_state=True):
        out = WoOut()
        if agent_id is None:
            rc = lib().wo_step(self._h, -1, None, C.byref(out))
        else:
            a = np.ascontiguousarray(action, dtype=np.float64).reshape(3)
            rc = lib().wo_step(self._h, int(agent_id), _dp(a), C.byref(out))
        if rc != 0:
            raise RuntimeError("wo_step failed rc=%d" % rc)
        return self._pack(out, with_state)

    def get_state(self, agent_id):
        s = np.empty((4, self.G, self.G), dtype=np.float64)
        lib().wo_get_state(self._h, int(agent_id), _dp(s))
        return s

    def nodes(self):
        E = np.empty(self.N); CS = np.empty(self.N); RR = np.empty(self.N)
        st = np.empty(self.N, dtype=np.int32); lv = np.empty(self.N, dtype=np.int32)
        lib().wo_peek_nodes(self._h, _dp(E), _dp(CS), _dp(RR), _ip(st), _ip(lv))
        return {"energy": E, "cs": CS, "rr": RR, "status": st, "level": lv}

    def mcs(self):
        a = np.empty((self.M, 14))
        lib().wo_peek_mcs(self._h, _dp(a))
        return {k: a[:, i].copy() for i, k in enumerate(MC_FIELDS)}

    def env_info(self):
        a = np.empty(14)
        lib().wo_peek_env(self._h, _dp(a))
        keys = ("xmin", "xmax", "ymin", "ymax", "nodes_density", "moving_time_max", "charging_time_max",
                "avg_nodes_agent", "now", "alive", "n_ticks", "n_hops", "n_events", "min_fitness")
        return dict(zip(keys, a.tolist()))

    def topology(self):
        deg = np.empty(self.N, dtype=np.int32); ncv = np.empty(self.N, dtype=np.int32); dr = np.empty(self.N, dtype=np.int32)
        ne = lib().wo_peek_topology(self._h, _ip(deg), _ip(ncv), _ip(dr))
        return {"degree": deg, "n_cover": ncv, "direct": dr, "n_edges": ne}

    def targets_active(self):
        t = np.empty(self.T, dtype=np.int32)
        lib().wo_peek_targets(self._h, _ip(t))
        return t
