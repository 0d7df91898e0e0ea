"""Diagnostic: G environment groups on G streams of one GPU (each its own handle), launches interleaved, vs one group."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = 4096; steps = 80
scs = [synth_scenario(e, 200, 200) for e in range(B)]
for groups in (1, 2, 4):
    n = B // groups
    streams = [torch.cuda.Stream() for _ in range(groups)]
    envs = []; reqs = []; gens = []
    for g in range(groups):
        with torch.cuda.stream(streams[g]):
            env = VecWRSN(scs[g * n:(g + 1) * n], None, 3, auto_reset=True, step_budget=1500)
            envs.append(env); reqs.append(env.reset()); gens.append(torch.Generator(device="cuda").manual_seed(g))
    def run(k):
        for _ in range(k):
            for g in range(groups):
                with torch.cuda.stream(streams[g]):
                    reqs[g] = envs[g].step(reqs[g]["agent_id"], torch.rand((n, 3), generator=gens[g], device="cuda", dtype=torch.float64))
    run(20); torch.cuda.synchronize()
    c0 = sum(e.counters()["env_steps"] for e in envs); t0 = time.perf_counter()
    run(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0; c1 = sum(e.counters()["env_steps"] for e in envs)
    print("groups %d: %.0f env-steps/s (%.3f ms per round of %d launches)" % (groups, (c1 - c0) / dt, 1e3 * dt / steps, groups))
    for e in envs: e.close()
