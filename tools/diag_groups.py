"""Diagnostic: throughput of the 4096-environment workload driven as G independent groups (own handle + HIP stream each)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = int(os.environ.get("WRSN_B", "4096")); K = int(os.environ.get("WRSN_K", "40")); W = 10
render = os.environ.get("WRSN_RENDER", "1") == "1"
scs = [synth_scenario(e, 200, 200) for e in range(B)]
for G in [int(x) for x in (sys.argv[1:] or ["1", "4", "8", "16", "32"])]:
    per = B // G
    streams = [torch.cuda.Stream() for _ in range(G)]
    envs, reqs = [], []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            envs.append(VecWRSN(scs[g * per:(g + 1) * per], None, 3, auto_reset=True, render=render, step_budget=int(os.environ.get("WRSN_BUDGET", "1500"))))
            reqs.append(envs[g].reset())
    gen = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.rand((W + K, B, 3), generator=gen, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    def run(k0, k1):
        for k in range(k0, k1):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    reqs[g] = envs[g].step(reqs[g]["agent_id"], acts[k, g * per:(g + 1) * per])
    run(0, W); torch.cuda.synchronize()
    c0 = sum(int(e.counters()["env_steps"]) for e in envs)
    t0 = time.perf_counter(); run(W, W + K); t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    c1 = sum(int(e.counters()["env_steps"]) for e in envs)
    print("G=%3d per=%5d: %.3f ms/round (cpu issue %.3f ms/round)  env-steps %d  -> %.0f env-steps/s" % (G, per, t / K * 1e3, t_cpu / K * 1e3, c1 - c0, (c1 - c0) / t), flush=True)
    for e in envs: e.close()
    del envs, reqs
