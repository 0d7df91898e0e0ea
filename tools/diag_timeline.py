"""Diagnostic: when do the waves of one step launch start and end (needs tools/libwrsn_hip_profile3.so, WRSN_PROFILE=3)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "tools", "libwrsn_hip_profile3.so")))
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = 4096
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "1250")), step_deadline_us=int(os.environ.get("WRSN_DEADLINE_US", "0")))
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
for k in range(25):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
    e0.record(); r = env.step(r["agent_id"], a); e1.record(); torch.cuda.synchronize()
    c = np.zeros((B * 25,), dtype=np.int64)
    _lib.check(env._h.lib, env._h.lib.wrsn_peek(env._h._h, 10, c.ctypes.data))
    p = c[:B * 24].reshape(B, 24)
    t0, t1 = p[:, 22].astype(np.float64), p[:, 23].astype(np.float64)
    base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0        # microseconds (100 MHz)
    st = r["status"].cpu().numpy()
    print("launch: %.0f us by events; waves span %.0f us; mean duration %.0f us; busy env (status 4): %d" % (e0.elapsed_time(e1) * 1e3, t1.max(), (t1 - t0).mean(), (st == 4).sum()))
    edges = np.linspace(0, t1.max(), 17)
    act = [int(((t0 < x) & (t1 > x)).sum()) for x in edges[:-1] + np.diff(edges) / 2]
    print("  active waves over time (16 bins): ", act)
    print("  start-time percentiles (us): p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(t0, [50, 90, 99, 100])))
    print("  end-time percentiles (us): p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % tuple(np.percentile(t1, [50, 90, 99, 99.9, 100])))
    long_ = (t1 - t0) > 0.6 * (t1 - t0).max()
    print("  long waves: %d, their start p50 %.0f p90 %.0f max %.0f" % (long_.sum(), *np.percentile(t0[long_], [50, 90, 100])))
    dur = t1 - t0
    slots = 2048
    print("  slot utilisation: %.1f %% of %d slots x span; wave durations us: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; sum %.0f slot-us" % (100.0 * dur.sum() / (slots * t1.max()), slots, *np.percentile(dur, [10, 50, 90, 99, 100]), dur.sum()))
    order = np.argsort(t0)
    late = order[slots:]                                   # waves that had to wait for a slot
    print("  second-round waves: %d, start p10 %.0f p50 %.0f p90 %.0f; their durations p50 %.0f p90 %.0f max %.0f; in flight after the launch %d" % (len(late), *np.percentile(t0[late], [10, 50, 90]), *np.percentile(dur[late], [50, 90, 100]), int((st[late] == 4).sum())))
    first = order[:slots]
    print("  first-round waves: durations p10 %.0f p50 %.0f p90 %.0f max %.0f; in flight after the launch %d" % (*np.percentile(dur[first], [10, 50, 90, 100]), int((st[first] == 4).sum())))
