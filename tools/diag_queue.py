"""Diagnostic: time-sliced launches (wrsn_set_step_deadline) -- rows per status and clock advance per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = 4096; D = int(os.environ.get("WRSN_DEADLINE_US", "400"))
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "0")), step_deadline_us=D)
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
c0 = env.counters()
for k in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
    e0.record(); r = env.step(r["agent_id"], a); e1.record(); torch.cuda.synchronize()
    st = r["status"]
    c1 = env.counters()
    print("launch %2d: %.0f us; status 0: %4d  3 (auto-reset): %4d  4 (in flight): %4d  other: %d; env-steps %d sim-seconds %d" % (
        k, e0.elapsed_time(e1) * 1e3, int((st == 0).sum()), int((st == 3).sum()), int((st == 4).sum()), int(((st != 0) & (st != 3) & (st != 4)).sum()),
        c1["env_steps"] - c0["env_steps"], c1["sim_seconds_total"] - c0["sim_seconds_total"]))
    c0 = c1
