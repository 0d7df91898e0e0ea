"""Diagnostic workload for profilers: the bench regime (4096 x 200 nodes x 3 chargers, budget 1250) without the observation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = int(os.environ.get("WRSN_B", "4096")); steps = int(os.environ.get("WRSN_STEPS", "60"))
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "1250")))
g = torch.Generator(device="cuda").manual_seed(1)
r = env.reset()
for k in range(steps):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
print("done", env.counters())
