"""Diagnostic: which process hops / services the slowest environments of a launch execute
(needs tools/libwrsn_hip_profile2.so built with WRSN_PROFILE_LEVEL=2)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "tools", "libwrsn_hip_profile2.so")))   # diagnostic override, tools only
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
NAMES = ["noff:stale_charging", "P_INIT", "MOVE_INIT", "MSTEP_INIT", "MSTEP_TIMEOUT", "MSTEP_DONE", "MOVE_DEADWAIT", "MOVE_DONE", "RECH_INIT",
         "RECH_TIMEOUT", "RECH_DONE", "CHG_INIT", "CSTEP_INIT", "CSTEP_TIMEOUT", "CSTEP_DONE", "CHG_DEADWAIT", "CHG_DONE", "P_DONE",
         "noff:second_process", "#grid_svc", "#grid_svc_stale_mover", "#ur_flag_evals", "#tie_grid_svc", "sum_entries"]
def prof(env):
    a = np.zeros((env.num_env * 25,), dtype=np.int64)
    _lib.check(env._h.lib, env._h.lib.wrsn_peek(env._h._h, 10, a.ctypes.data))
    return a[:env.num_env * 24].reshape(env.num_env, 24).copy(), a[env.num_env * 24:].copy()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scs = [synth_scenario(e, 200, 200) for e in range(B)]
env = VecWRSN(scs, None, 3, auto_reset=True, render=False)
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
for k in range(20):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
p0, k0 = prof(env)
for k in range(8):
    a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
    r = env.step(r["agent_id"], a); torch.cuda.synchronize()
    p1, k1 = prof(env)
    d = p1 - p0; kd = k1 - k0
    ev = d[:, :19].sum(1)
    print("launch %d: events/env mean %.1f p99 %.0f max %d" % (k, ev.mean(), np.percentile(ev, 99), ev.max()))
    for e in np.argsort(-kd)[:4]:
        print("  env %d cycles %d: " % (e, kd[e]) + "  ".join("%s=%d" % (n, d[e, i]) for i, n in enumerate(NAMES) if d[e, i]))
    p0, k0 = p1, k1
