"""Soak run: many launches of the benchmark workload, checking status / rewards / clocks for anything abnormal."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = int(os.environ.get("WRSN_B", "4096")); K = int(os.environ.get("WRSN_K", "3000")); budget = int(os.environ.get("WRSN_BUDGET", "1250"))
NN = int(os.environ.get("WRSN_N", "200")); MM = int(os.environ.get("WRSN_M", "3"))
env = VecWRSN([synth_scenario(7000 + e, NN, NN) for e in range(B)], None, MM, auto_reset=True, step_budget=budget, step_deadline_us=int(os.environ.get("WRSN_DEADLINE_US", "0")), reuse_obs=True)
g = torch.Generator(device="cuda").manual_seed(11)
r = env.reset()
bad = torch.zeros((), dtype=torch.int64, device="cuda"); nan = torch.zeros((), dtype=torch.int64, device="cuda")
busy_run = torch.zeros(B, dtype=torch.int32, device="cuda"); worst = torch.zeros((), dtype=torch.int32, device="cuda")
t0 = time.time()
for k in range(K):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
    st = r["status"]
    bad += (st < 0).sum(); nan += (~torch.isfinite(r["reward"])).sum() + (~torch.isfinite(r["now"])).sum()
    busy_run = torch.where(st == 4, busy_run + 1, torch.zeros_like(busy_run)); worst = torch.maximum(worst, busy_run.max())
    if k % 500 == 499:
        torch.cuda.synchronize(); print("launch %d: negative status %d, non-finite %d, longest in-flight run %d launches, %.1f s" % (k + 1, int(bad), int(nan), int(worst), time.time() - t0), flush=True)
torch.cuda.synchronize()
c = env.counters(); tab = env.rollout_table().cpu()
print("env-steps %d, episodes %d, mean lifetime %.1f s, obs finite %s" % (c["env_steps"], int(tab[:, MM].sum()), float(tab[:, MM + 1].sum() / max(1.0, float(tab[:, MM].sum()))), bool(torch.isfinite(r["state"]).all())))
assert int(bad) == 0 and int(nan) == 0
print("soak ok")
