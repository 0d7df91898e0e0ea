"""ctypes binding of libwrsn_hip.so (C-ABI: include/wrsn_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950) and lives next to
its sources in `csrc/`.  There is no CPU fallback: `load()` raises when the shared object is missing, and
`wrsn_create` fails with WRSN_ERR_NO_DEVICE when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libwrsn_hip.so")

WRSN_OK = 0
ERR_NAMES = {0: "WRSN_OK", -1: "WRSN_ERR_ARG", -2: "WRSN_ERR_HIP", -3: "WRSN_ERR_NO_DEVICE",
             -4: "WRSN_ERR_CAPACITY", -5: "WRSN_ERR_STATE"}

PEEK_NODE_ENERGY, PEEK_NODE_CS, PEEK_NODE_RR, PEEK_NODE_STATUS, PEEK_NODE_LEVEL = 0, 1, 2, 3, 4
PEEK_MC, PEEK_ENV, PEEK_NODE_DEGREE, PEEK_NODE_NCOVER, PEEK_NODE_DIRECT = 5, 6, 7, 8, 9
PEEK_TARGETS_ACTIVE = 11
MC_FIELDS = ("loc_x", "loc_y", "energy", "status", "type_charging", "cur_x", "cur_y", "cur_t", "n_conn",
             "excl", "prev_minfit", "act0", "act1", "act2", "_r0", "_r1")
ENV_FIELDS = ("xmin", "xmax", "ymin", "ymax", "nodes_density", "moving_time_max", "charging_time_max",
              "avg_nodes_agent", "now", "alive", "n_ticks", "n_exact", "n_events", "min_fitness", "n_edges", "n_cover")

# every entry point include/wrsn_hip.h declares
EXPORTS = ("wrsn_create", "wrsn_destroy", "wrsn_set_stream", "wrsn_set_scenario", "wrsn_reset", "wrsn_step",
           "wrsn_set_step_budget", "wrsn_set_step_deadline", "wrsn_density_action", "wrsn_rollout_table", "wrsn_rollout_record", "wrsn_rollout_collect", "wrsn_render", "wrsn_set_obs_reuse", "wrsn_set_timing", "wrsn_kernel_times", "wrsn_peek", "wrsn_sync", "wrsn_counters", "wrsn_synth_network",
           "wrsn_last_error",
           "wrsn_version")


class WrsnCfg(C.Structure):
    _fields_ = [("n_env", C.c_int32), ("n_node", C.c_int32), ("n_target", C.c_int32), ("n_mc", C.c_int32),
                ("map_size", C.c_int32), ("device", C.c_int32), ("max_degree", C.c_int32), ("max_cover", C.c_int32),
                ("warm_up_time", C.c_double)]


class WrsnNodeSpec(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("capacity", "threshold", "com_range", "sen_range", "prob_gp",
                                          "package_size", "er", "et", "efs", "emp", "max_time")]


class WrsnMcSpec(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("capacity", "threshold", "velocity", "pm", "charging_range", "alpha",
                                          "beta", "epsilon")]


class WrsnStepOut(C.Structure):
    _fields_ = [("agent_id", C.c_void_p), ("reward", C.c_void_p), ("terminal", C.c_void_p), ("now", C.c_void_p),
                ("obs", C.c_void_p), ("status", C.c_void_p)]


class WrsnTransitionBuffers(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("action_elems", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("pend_state", "pend_action", "pend_logp", "pend_valid", "state", "action", "next_state",
                                          "reward", "logp", "now", "env", "count")]


class WrsnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "WRSN_ERR"), code, msg))
        self.code = code


def bind(lib):
    """Declare the signatures of include/wrsn_hip.h on a loaded shared object."""
    vp = C.c_void_p
    lib.wrsn_create.argtypes = [C.POINTER(WrsnCfg), C.POINTER(vp)]
    lib.wrsn_create.restype = C.c_int
    lib.wrsn_destroy.argtypes = [vp]
    lib.wrsn_destroy.restype = None
    lib.wrsn_set_stream.argtypes = [vp, vp]
    lib.wrsn_set_stream.restype = C.c_int
    lib.wrsn_set_scenario.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(WrsnNodeSpec), C.c_int32,
                                      C.POINTER(WrsnMcSpec), C.c_int32]
    lib.wrsn_set_scenario.restype = C.c_int
    lib.wrsn_reset.argtypes = [vp, vp, C.POINTER(WrsnStepOut)]
    lib.wrsn_reset.restype = C.c_int
    lib.wrsn_step.argtypes = [vp, vp, vp, C.c_int32, C.POINTER(WrsnStepOut)]
    lib.wrsn_step.restype = C.c_int
    lib.wrsn_set_step_budget.argtypes = [vp, C.c_int32]
    lib.wrsn_set_step_budget.restype = C.c_int
    lib.wrsn_set_step_deadline.argtypes = [vp, C.c_int32]
    lib.wrsn_set_step_deadline.restype = C.c_int
    lib.wrsn_density_action.argtypes = [vp, vp, vp, vp]
    lib.wrsn_density_action.restype = C.c_int
    lib.wrsn_rollout_table.argtypes = [vp, vp, C.c_int32]
    lib.wrsn_rollout_table.restype = C.c_int
    lib.wrsn_rollout_record.argtypes = [vp, C.POINTER(WrsnTransitionBuffers), vp, vp, vp, vp]
    lib.wrsn_rollout_record.restype = C.c_int
    lib.wrsn_rollout_collect.argtypes = [vp, C.POINTER(WrsnTransitionBuffers), C.POINTER(WrsnStepOut)]
    lib.wrsn_rollout_collect.restype = C.c_int
    lib.wrsn_set_obs_reuse.argtypes = [vp, C.c_int32]
    lib.wrsn_set_obs_reuse.restype = C.c_int
    lib.wrsn_set_timing.argtypes = [vp, C.c_int32]
    lib.wrsn_set_timing.restype = C.c_int
    lib.wrsn_kernel_times.argtypes = [vp, vp]
    lib.wrsn_kernel_times.restype = C.c_int
    lib.wrsn_render.argtypes = [vp, vp, vp]
    lib.wrsn_render.restype = C.c_int
    lib.wrsn_peek.argtypes = [vp, C.c_int32, vp]
    lib.wrsn_peek.restype = C.c_int
    lib.wrsn_sync.argtypes = [vp]
    lib.wrsn_sync.restype = C.c_int
    lib.wrsn_counters.argtypes = [vp, vp]
    lib.wrsn_counters.restype = C.c_int
    lib.wrsn_synth_network.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    lib.wrsn_synth_network.restype = C.c_int
    lib.wrsn_last_error.argtypes = []
    lib.wrsn_last_error.restype = C.c_char_p
    lib.wrsn_version.argtypes = []
    lib.wrsn_version.restype = C.c_char_p
    return lib


_lib = None


def load():
    """Load the in-tree HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own copy of the HIP runtime (same SONAME as /opt/rocm's).  Whichever copy is mapped first serves every
        # later dlopen; if this library came first, torch would bring a SECOND runtime into the process and the one bound here
        # would find no device ("no ROCm-capable device is detected").  Import torch first so that there is exactly one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        alt = os.environ.get("WRSN_HIP_LIB")                   # diagnostic: another BUILD OF THE SAME HIP library (A/B timing runs)
        if alt:
            _lib = bind(C.CDLL(alt))
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  multi_agent_rl_wrsn_amd has no CPU fallback." % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def check(lib, rc):
    if rc != WRSN_OK:
        msg = lib.wrsn_last_error()
        raise WrsnError(rc, msg.decode() if msg else "")


def make_node_spec(node_spec, max_time):
    s = WrsnNodeSpec()
    for k in ("capacity", "threshold", "com_range", "sen_range", "prob_gp", "package_size", "er", "et", "efs", "emp"):
        setattr(s, k, float(node_spec[k]))
    s.max_time = float(max_time)
    return s


def make_mc_spec(mc_spec):
    s = WrsnMcSpec()
    for k in ("capacity", "threshold", "velocity", "pm", "charging_range", "alpha", "beta", "epsilon"):
        setattr(s, k, float(mc_spec[k]))
    return s


class RawHandle:
    """Thin owner of one wrsn_t*.  Array arguments are raw addresses (ints): device pointers for the HIP
    library.  Used by VecWRSN (torch tensors) and, with the emulated library, by the CPU logic tests."""

    def __init__(self, lib, n_env, n_node, n_target, n_mc, map_size=100, warm_up_time=100.0, device=0,
                 max_degree=0, max_cover=0):
        self.lib = lib
        self.cfg = WrsnCfg(int(n_env), int(n_node), int(n_target), int(n_mc), int(map_size), int(device),
                           int(max_degree), int(max_cover), float(warm_up_time))
        self._h = C.c_void_p()
        check(lib, lib.wrsn_create(C.byref(self.cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.wrsn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(self.lib, self.lib.wrsn_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_scenarios(self, scenarios, mc_spec, env0=0):
        """scenarios: list of Scenario (len = number of environments to set starting at env0)."""
        import numpy as np
        n = len(scenarios)
        N, T = self.cfg.n_node, self.cfg.n_target
        node_xy = np.zeros((n, N, 2)); target_xy = np.zeros((n, T, 2)); bs = np.zeros((n, 2))
        nn = np.zeros(n, dtype=np.int32); nt = np.zeros(n, dtype=np.int32)
        specs = (WrsnNodeSpec * n)()
        for e, sc in enumerate(scenarios):
            if sc.n_node > N or sc.n_target > T:
                raise ValueError("scenario %d has %d nodes / %d targets, handle was created for %d / %d" %
                                 (e, sc.n_node, sc.n_target, N, T))
            node_xy[e, :sc.n_node] = sc.node_xy; target_xy[e, :sc.n_target] = sc.target_xy; bs[e] = sc.bs_xy
            nn[e], nt[e] = sc.n_node, sc.n_target
            specs[e] = make_node_spec(sc.node_spec, sc.max_time)
        mcs = make_mc_spec(mc_spec)
        check(self.lib, self.lib.wrsn_set_scenario(self._h, int(env0), n, node_xy.ctypes.data, target_xy.ctypes.data,
                                                   bs.ctypes.data, nn.ctypes.data, nt.ctypes.data, specs, 1,
                                                   C.byref(mcs), 0))

    @staticmethod
    def _out(agent_id=0, reward=0, terminal=0, now=0, obs=0, status=0):
        return WrsnStepOut(agent_id or None, reward or None, terminal or None, now or None, obs or None, status or None)

    def reset(self, mask_ptr=0, **out_ptrs):
        o = self._out(**out_ptrs)
        check(self.lib, self.lib.wrsn_reset(self._h, C.c_void_p(mask_ptr or None), C.byref(o)))

    def step(self, agent_ptr, action_ptr, auto_reset=False, **out_ptrs):
        o = self._out(**out_ptrs)
        check(self.lib, self.lib.wrsn_step(self._h, C.c_void_p(agent_ptr), C.c_void_p(action_ptr), int(bool(auto_reset)), C.byref(o)))

    def set_step_budget(self, work_units):
        check(self.lib, self.lib.wrsn_set_step_budget(self._h, int(work_units)))

    def set_step_deadline(self, microseconds):
        check(self.lib, self.lib.wrsn_set_step_deadline(self._h, int(microseconds)))

    def density_action(self, agent_ptr, dmap_ptr, action_ptr):
        check(self.lib, self.lib.wrsn_density_action(self._h, C.c_void_p(agent_ptr), C.c_void_p(dmap_ptr), C.c_void_p(action_ptr)))

    def rollout_table(self, dst_ptr, zero_after=False):
        check(self.lib, self.lib.wrsn_rollout_table(self._h, C.c_void_p(dst_ptr), 1 if zero_after else 0))

    def rollout_record(self, buffers, agent_ptr, action_ptr, logp_ptr, obs_ptr):
        check(self.lib, self.lib.wrsn_rollout_record(self._h, C.byref(buffers), C.c_void_p(agent_ptr), C.c_void_p(action_ptr),
                                                     C.c_void_p(logp_ptr), C.c_void_p(obs_ptr)))

    def rollout_collect(self, buffers, **out_ptrs):
        o = self._out(**out_ptrs)
        check(self.lib, self.lib.wrsn_rollout_collect(self._h, C.byref(buffers), C.byref(o)))

    def set_obs_reuse(self, on):
        check(self.lib, self.lib.wrsn_set_obs_reuse(self._h, 1 if on else 0))

    def set_timing(self, on):
        check(self.lib, self.lib.wrsn_set_timing(self._h, 1 if on else 0))

    def kernel_times(self):
        import numpy as np
        a = np.zeros(4, dtype=np.float32)
        check(self.lib, self.lib.wrsn_kernel_times(self._h, a.ctypes.data))
        return {"order_ms": float(a[0]), "step_ms": float(a[1]), "continuation_ms": float(a[2]), "obs_ms": float(a[3])}

    def render(self, agent_ptr, obs_ptr):
        check(self.lib, self.lib.wrsn_render(self._h, C.c_void_p(agent_ptr), C.c_void_p(obs_ptr)))

    def sync(self):
        check(self.lib, self.lib.wrsn_sync(self._h))

    def peek(self, what):
        import numpy as np
        B, N, M = self.cfg.n_env, self.cfg.n_node, self.cfg.n_mc
        if what in (PEEK_NODE_ENERGY, PEEK_NODE_CS, PEEK_NODE_RR):
            a = np.empty((B, N), dtype=np.float64)
        elif what in (PEEK_NODE_STATUS, PEEK_NODE_LEVEL, PEEK_NODE_DEGREE, PEEK_NODE_NCOVER, PEEK_NODE_DIRECT):
            a = np.empty((B, N), dtype=np.int32)
        elif what == PEEK_TARGETS_ACTIVE:
            a = np.empty((B, self.cfg.n_target), dtype=np.int32)
        elif what == PEEK_MC:
            a = np.empty((B, M, 16), dtype=np.float64)
        elif what == PEEK_ENV:
            a = np.empty((B, 16), dtype=np.float64)
        else:
            raise ValueError("unknown peek selector %r" % (what,))
        check(self.lib, self.lib.wrsn_peek(self._h, int(what), a.ctypes.data))
        return a

    def counters(self):
        import numpy as np
        a = np.zeros(8, dtype=np.int64)
        check(self.lib, self.lib.wrsn_counters(self._h, a.ctypes.data))
        return {"ticks": int(a[0]), "exact_ticks": int(a[1]), "events": int(a[2]), "env_steps": int(a[3]),
                "sim_seconds_total": int(a[4]), "zero_time_steps": int(a[5])}
