"""Diagnostic: breakdown of the scalar event processor / service loop (needs tools/libwrsn_hip_profile4.so, WRSN_PROFILE=4)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "tools", "libwrsn_hip_profile4.so")))   # diagnostic override, tools only
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
NAMES = ["scalar_run", "#loop_iterations", "event_scan", "#scans", "ur_flags+build", "#ur_builds", "thread_fire", "#fires", "pend_tail", "ff_sync(tie)",
         "decide", "run:post+barrier+mailbox(incl scalar_run)", "svc:precheck", "svc:conn_build", "svc:grid", "min_fitness", "load", "store", "prologue+bind", "-",
         "#services", "#grid_services", "-", "-"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "1250")))
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
def prof():
    a = np.zeros((B * 25,), dtype=np.int64)
    _lib.check(env._h.lib, env._h.lib.wrsn_peek(env._h._h, 10, a.ctypes.data))
    return a[:B * 24].reshape(B, 24).copy(), a[B * 24:].copy()
for k in range(20):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
p0, k0 = prof(); c0 = env.counters()
L = 10
for k in range(L):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
p1, k1 = prof(); c1 = env.counters()
d = (p1 - p0) / L; kd = (k1 - k0) / L
print("per environment per launch: kernel cycles mean %.0f; completed steps per launch %.0f" % (kd.mean(), (c1["env_steps"] - c0["env_steps"]) / L))
for i, n in enumerate(NAMES):
    if n != "-": print("%-46s %12.1f" % (n, d[:, i].mean()))
