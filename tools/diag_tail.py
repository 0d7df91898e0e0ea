"""Diagnostic: phase breakdown of the slowest environment of every launch (needs tools/libwrsn_hip_profile.so)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "tools", "libwrsn_hip_profile.so")))   # diagnostic override, tools only
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
from diag_phases import NAMES, prof  # noqa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scs = [synth_scenario(e, 200, 200) for e in range(B)]
env = VecWRSN(scs, None, 3, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "0")))
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
for k in range(20):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
p0, k0 = prof(env)
info0 = env.env_info()
for k in range(8):
    a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
    r = env.step(r["agent_id"], a); torch.cuda.synchronize()
    p1, k1 = prof(env); info1 = env.env_info()
    d = p1 - p0; kd = k1 - k0
    w = np.argsort(-kd)[:3]
    print("launch %d: kernel cycles mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (k, kd.mean(), np.percentile(kd, 50), np.percentile(kd, 90), np.percentile(kd, 99), kd.max()))
    print("  MEAN    " + "  ".join("%s=%d" % (n, d[:, i].mean()) for i, n in enumerate(NAMES) if d[:, i].any()))
    for e in w:
        print("  env %d cycles %d  dticks %.0f dexact %.0f devents %.0f status %d" % (e, kd[e], info1["n_ticks"][e] - info0["n_ticks"][e], info1["n_exact"][e] - info0["n_exact"][e], info1["n_events"][e] - info0["n_events"][e], int(r["status"][e])))
        print("    " + "  ".join("%s=%d" % (n, d[e, i]) for i, n in enumerate(NAMES) if d[e, i]))
    p0, k0, info0 = p1, k1, info1
