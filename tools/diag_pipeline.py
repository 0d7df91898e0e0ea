"""Diagnostic: the bench workload as G groups of environments, each with its own handle and HIP stream, stepped round-robin.
The observation kernel of one group then runs beside the step kernel of another, and the tail of a step launch is
filled by the next group's waves.  python tools/diag_pipeline.py <B> <groups> <budget> <steps> [lean]
lean: actions pre-generated, the request's agent_id tensor handed straight back (no torch op per step: host cost = 4 launches)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
lean = len(sys.argv) > 5 and sys.argv[5] == "lean"
dev = torch.device("cuda:0")
scs = [synth_scenario(e, 200, 200) for e in range(B)]
per = B // G
streams = [torch.cuda.Stream(dev) for _ in range(G)]
envs, gens, reqs, acts = [], [], [], []
for g in range(G):
    with torch.cuda.stream(streams[g]):
        envs.append(VecWRSN(scs[g * per:(g + 1) * per], None, 3, device=str(dev), auto_reset=True, step_budget=budget, reuse_obs=True))
        gens.append(torch.Generator(device=dev).manual_seed(7 + g))
        reqs.append(envs[g].reset())
        acts.append(torch.rand((steps + 20, per, 3), generator=gens[g], device=dev, dtype=torch.float64))
        envs[g]._bind_stream()
torch.cuda.synchronize(dev)

def one_pass(k):
    for g in range(G):
        if lean:
            e = envs[g]
            e._h.step(e.agent_id.data_ptr(), acts[g][k].data_ptr(), True, **e._out_ptrs())
        else:
            with torch.cuda.stream(streams[g]):
                reqs[g] = envs[g].step(reqs[g]["agent_id"], acts[g][k])

for k in range(20):
    one_pass(k)
torch.cuda.synchronize(dev)
c0 = sum(e.counters()["env_steps"] for e in envs)
t0 = time.perf_counter()
for k in range(steps):
    one_pass(20 + k)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(dev)
el = time.perf_counter() - t0
c1 = sum(e.counters()["env_steps"] for e in envs)
print("B=%d groups=%d budget=%d lean=%d: %.3f ms per pass (host enqueue %.3f), %.0f env-steps per pass, %.3f M env-steps/s" %
      (B, G, budget, lean, 1e3 * el / steps, 1e3 * t_host / steps, (c1 - c0) / steps, (c1 - c0) / el / 1e6), flush=True)
