"""Diagnostic: host-side time per VecWRSN.step call segment (no GPU sync inside the loop)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario, RolloutStats
B = 4096
scs = [synth_scenario(e, 200, 200) for e in range(B)]
env = VecWRSN(scs, None, 3, auto_reset=True)
dev = env.device
gen = torch.Generator(device=dev).manual_seed(0)
stats = RolloutStats(B, 3, dev)
r = env.reset()
for _ in range(5):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=gen, device=dev, dtype=torch.float64))
torch.cuda.synchronize()
T = {"rand": 0.0, "step": 0.0, "stats": 0.0}
t_all = time.perf_counter()
for _ in range(40):
    t0 = time.perf_counter(); a = torch.rand((B, 3), generator=gen, device=dev, dtype=torch.float64)
    t1 = time.perf_counter(); r = env.step(r["agent_id"], a)
    t2 = time.perf_counter(); stats.update(r["agent_id"], r["reward"], r["terminal"], r["now"])
    t3 = time.perf_counter()
    T["rand"] += t1 - t0; T["step"] += t2 - t1; T["stats"] += t3 - t2
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
print("host enqueue per step: %.3f ms (rand %.3f, step %.3f, stats %.3f); wall per step incl. GPU: %.3f ms" % (
    1e3 * t_enq / 40, 1e3 * T["rand"] / 40, 1e3 * T["step"] / 40, 1e3 * T["stats"] / 40, 1e3 * t_tot / 40))
