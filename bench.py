"""bench.py -- env-steps/s of the WRSN environment step path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): 4096 independent environments per GPU x 200 nodes x 200 targets x 3 mobile
chargers, seeded synthetic networks, random policy actions ~U[0,1)^3, auto-reset on terminal.
A "step" is one VecWRSN.step launch over the whole batch.  With the default step budget an environment whose WRSN.step
is not finished when its share of the launch is used up reports "still running" and goes on in the next launch (the
duration of a WRSN.step is heavy-tailed; see include/wrsn_hip.h, wrsn_set_step_budget): `value` counts COMPLETED
WRSN.step() calls only (each with its 4x100x100 observation) -- auto-resets and unfinished steps are not counted.
The same workload with blocking steps (budget 0, every launch waits for its slowest environment) is measured too and
reported under "blocking".
Environments are independent, so ranks shard them with no data-path collective (weak scaling); the only exchange is
one all-gather of the rollout returns table after the timed region.

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process starts N rank processes of itself -- before it
touches torch or the GPU -- with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR 127.0.0.1),
forwards rank 0's JSON line and exits with the worst return code.  The line carries `ranks`, the size of the RCCL
group the ranks really formed (counted with an all-reduce), next to `n_gpus`.

Next to env-steps/s the line reports `sim_ticks_per_s` (simulated seconds x environments per wall
second), `mean_ticks_per_env_step` and `zero_time_step_share` (completed WRSN.step calls that returned at the instant
they were called: the bookkeeping returns at t = 100 after every reset).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(N, T, M, G):
    """compulsory HBM bytes per env-step with state resident on chip during the step."""
    physics = 112 * N + 8 * T + 104 * M + 28
    obs = 16 * G * G
    return physics, obs


def cpu_baseline(scenarios, M, seconds, threads):
    """The oracle (plain-C float64 restatement of the reference, oracle/) timed on the host cores: a bounded sample of
    the same workload (same networks, same action distribution), one environment per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from multi_agent_rl_wrsn_amd import DEFAULT_MC_SPEC
    from wrsn_oracle import OracleWRSN, lib
    lib()
    deadline = time.time() + seconds

    def work(i):
        s = scenarios[i % len(scenarios)]
        o = OracleWRSN(s.node_xy, s.target_xy, s.bs_xy, s.node_spec, DEFAULT_MC_SPEC, s.max_time, M)
        rng = np.random.RandomState(i)
        r = o.reset(with_state=True)
        n = 0
        while time.time() < deadline:
            if r["terminal"]:
                r = o.reset(with_state=True)
                continue
            r = o.step(r["agent_id"], rng.rand(3), with_state=True)     # includes get_state, like WRSN.step
            n += 1
        return n

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        counts = list(ex.map(work, range(threads)))
    dt = time.time() - t0
    return sum(counts) / dt, sum(counts), dt


def measure(torch, dist, dev, scenarios, M, G, rank, seed, budget, steps, warmup, kernel_steps, deadline_us=0, min_seconds=0.0):
    """One timed run of `steps` VecWRSN.step launches on this rank's shard; returns the local numbers."""
    from multi_agent_rl_wrsn_amd import RolloutStats, VecWRSN
    B = len(scenarios)
    t_set = time.time()
    env = VecWRSN(scenarios, None, M, map_size=G, device=str(dev), auto_reset=True, step_budget=budget, step_deadline_us=deadline_us,
                  reuse_obs=True)                             # the random policy never writes into the state tensor
    env.synchronize()
    t_set = time.time() - t_set
    gen = torch.Generator(device=dev).manual_seed(seed * 7919 + rank)

    def policy():
        return torch.rand((B, 3), generator=gen, device=dev, dtype=torch.float64)

    r = env.reset()
    for _ in range(max(1, warmup)):                            # untimed: also loads every torch kernel the timed loop uses
        r = env.step(r["agent_id"], policy())
    env.rollout_table(zero_after=True)                          # returns / episode counters are accumulated by the step kernel itself
    torch.cuda.synchronize(dev)
    c0 = env.counters()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = env.step(r["agent_id"], policy())
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    c1 = env.counters()
    res = {"elapsed": elapsed, "env_steps": c1["env_steps"] - c0["env_steps"], "sim_seconds": c1["sim_seconds_total"] - c0["sim_seconds_total"],
           "zero_steps": c1["zero_time_steps"] - c0["zero_time_steps"], "exact_ticks": c1["exact_ticks"],
           "table": RolloutStats.gather_table(env.rollout_table()), "t_set": t_set}   # the path's one exchange step (RCCL all-gather)
    # ---- a timed region of K steps that lasts less than min_seconds is too short for samplers around the run (the driver passes
    #      --steps 20: ~12 ms): a second region of as many steps as min_seconds needs is timed the same way and reported beside it
    res["ext"] = None
    el_all = elapsed
    if dist and min_seconds > 0:                               # one decision for the whole job (the ranks' clocks differ by microseconds:
        et_ = torch.tensor([elapsed], dtype=torch.float64, device=dev); dist.all_reduce(et_, op=dist.ReduceOp.MAX); el_all = float(et_[0])   # a rank alone in a collective would hang)
    if min_seconds > 0 and el_all < min_seconds:
        k2 = int(min(100000, max(steps + 1, steps * min_seconds / max(el_all, 1e-6) * 1.2)))
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k2):
            r = env.step(r["agent_id"], policy())
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        el2 = time.perf_counter() - t0
        c2 = env.counters()
        res["ext"] = {"steps": k2, "elapsed": el2, "env_steps": c2["env_steps"] - c1["env_steps"], "zero_steps": c2["zero_time_steps"] - c1["zero_time_steps"]}

    # ---- per-kernel timing pass: HIP events recorded by the library itself on the stream the kernels are launched on
    #      (wrsn_set_timing / wrsn_kernel_times): launch-order kernels, step kernel and observation kernel separately
    t_env = t_obs = t_ord = 0.0
    ks = max(1, kernel_steps)
    env._h.set_timing(True)
    steps_before = env.counters()["env_steps"]
    for _ in range(ks):
        r = env.step(r["agent_id"], policy())
        kt = env._h.kernel_times()
        t_env += (kt["step_ms"] + kt["continuation_ms"]) * 1e-3; t_obs += kt["obs_ms"] * 1e-3; t_ord += kt["order_ms"] * 1e-3
    env._h.set_timing(False)
    res["units"] = (env.counters()["env_steps"] - steps_before) / ks    # env-steps one launch completes (auto-resets excluded)
    res["env_launch"] = t_env / ks; res["obs_launch"] = t_obs / ks; res["order_launch"] = t_ord / ks
    res["mean_episode_seconds"] = float(env.env_info()["n_ticks"].mean())
    env.close()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, before this process touches the GPU
    (multi_agent_rl_wrsn_amd.sharding.launch_ranks; importing the package loads neither torch's GPU runtime nor the HIP library)."""
    from multi_agent_rl_wrsn_amd.sharding import launch_ranks
    raise SystemExit(launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--targets", type=int, default=200)
    ap.add_argument("--mcs", type=int, default=3)
    ap.add_argument("--map-size", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--kernel-steps", type=int, default=20, help="steps of the per-kernel timing pass")
    ap.add_argument("--step-budget", type=int, default=1250,
                    help="work units one launch may spend per environment (VecWRSN step_budget); 0 = blocking steps: every "
                         "launch waits for its slowest WRSN.step")
    ap.add_argument("--step-deadline-us", type=int, default=0,
                    help="common deadline of a launch in microseconds after its first wave started (VecWRSN step_deadline_us; 0 = none): "
                         "waves still running then stop at the next item boundary like waves out of budget")
    ap.add_argument("--min-seconds", type=float, default=0.1,
                    help="when the timed region of --steps launches is shorter than this, a second region long enough is timed too and reported as `extended` (0 = off)")
    ap.add_argument("--no-blocking-run", action="store_true", help="skip the additional blocking-mode (step_budget 0) measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                                  # never returns

    import torch
    from multi_agent_rl_wrsn_amd import init_distributed, synth_scenario
    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch.distributed if world > 1 else None
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ranks = 1
    if dist:                                                   # how many ranks the RCCL group really has
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)
        ranks = int(one[0])
        if ranks != args.gpus or dist.get_backend() != "nccl":
            raise SystemExit("process group has %d ranks on backend %s, expected %d on nccl" % (ranks, dist.get_backend(), args.gpus))

    B, N, T, M, G = args.envs, args.nodes, args.targets, args.mcs, args.map_size
    t_gen = time.time()
    env0 = rank * B                                            # global environment ids of this shard
    scenarios = [synth_scenario(args.seed + env0 + e, N, T) for e in range(B)]
    t_gen = time.time() - t_gen

    def reduced(res):
        ext = res.get("ext") or {"elapsed": 0.0, "env_steps": 0, "zero_steps": 0}
        el = torch.tensor([res["elapsed"], ext["elapsed"]], dtype=torch.float64, device=dev)
        cnt = torch.tensor([res["env_steps"], res["sim_seconds"], res["zero_steps"], ext["env_steps"], ext["zero_steps"]], dtype=torch.float64, device=dev)
        if dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)           # slowest rank
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)          # whole job
        res["sim_seconds_all"], res["zero_steps_all"] = float(cnt[1]), float(cnt[2])
        if res.get("ext"):
            res["ext"]["elapsed_all"], res["ext"]["env_steps_all"], res["ext"]["zero_steps_all"] = float(el[1]), float(cnt[3]), float(cnt[4])
        return float(el[0]), float(cnt[0])

    main_res = measure(torch, dist, dev, scenarios, M, G, rank, args.seed, args.step_budget, args.steps, args.warmup, args.kernel_steps, args.step_deadline_us, args.min_seconds)
    elapsed, env_steps = reduced(main_res)
    blocking = None
    if args.step_budget > 0 and not args.no_blocking_run:
        bsteps = max(10, args.steps // 2)
        bres = measure(torch, dist, dev, scenarios, M, G, rank, args.seed, 0, bsteps, max(5, args.warmup // 2), max(2, args.kernel_steps // 2))
        bel, bcnt = reduced(bres)
        blocking = {"value": bcnt / bel, "unit": "env-steps/s", "steps": bsteps, "ms_per_step": 1e3 * bel / bsteps,
                    "sim_ticks_per_s": bres["sim_seconds_all"] / bel, "mean_ticks_per_env_step": bres["sim_seconds_all"] / max(1.0, bcnt),
                    "zero_time_step_share": bres["zero_steps_all"] / max(1.0, bcnt),
                    "value_time_advancing": (bcnt - bres["zero_steps_all"]) / bel,
                    "kernels": {"wrsn_step_kernel_ms": 1e3 * bres["env_launch"], "wrsn_obs_kernel_ms": 1e3 * bres["obs_launch"],
                                "launch_order_kernels_ms": 1e3 * bres["order_launch"], "env_steps_per_launch": bres["units"]},
                    "note": "step_budget 0: every launch runs each WRSN.step to its end and waits for the slowest environment"}

    phys_b, obs_b = algorithmic_bytes(N, T, M, G)
    env_launch, obs_launch, units = main_res["env_launch"], main_res["obs_launch"], main_res["units"]
    if env_launch >= obs_launch:
        dom, dur, per_unit = "wrsn_step_kernel", env_launch, phys_b
    else:
        dom, dur, per_unit = "wrsn_obs_kernel", obs_launch, obs_b
    achieved = per_unit * units / dur / 1e9
    peak = 8000.0
    # HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (profiles/), attached
    # only when they were taken on THIS configuration (the file names it); null otherwise
    traffic = None; traffic_src = None; rocprof_us = None
    this_cfg = {"envs_per_gpu": B, "nodes": N, "targets": T, "chargers": M, "map_size": G, "step_budget": args.step_budget}
    try:
        for tfile in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
            if tj.get("config") == this_cfg and dom in tj.get("kernels", {}):
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]; traffic_src = "profiles/" + tfile
                rocprof_us = tj["kernels"][dom].get("rocprof_avg_us")     # average duration of that kernel in the --kernel-trace --stats pass of the same command
                break
    except Exception:
        traffic = None

    if rank == 0:
        value = env_steps / elapsed
        out = {
            "metric": "env-steps/sec (whole node), 4096 envs x 200 nodes, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "ranks": ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 (physics) / f32 (observation)", "data": "synthetic",
            "config": {"workload": "%d envs/GPU x %d nodes x %d targets x %d MC, random policy U[0,1)^3, auto-reset, 4x%dx%d observation" % (B, N, T, M, G, G),
                       "step_budget": args.step_budget, "step_deadline_us": args.step_deadline_us, "envs_per_gpu": B, "nodes": N, "targets": T, "chargers": M, "map_size": G,
                       "parallelism": "env-shard x%d" % world},
            "env_steps_timed": env_steps, "sim_ticks_per_s": main_res["sim_seconds_all"] / elapsed,
            "mean_ticks_per_env_step": main_res["sim_seconds_all"] / max(1.0, env_steps),
            "zero_time_step_share": main_res["zero_steps_all"] / max(1.0, env_steps),
            "mean_episode_seconds_so_far": main_res["mean_episode_seconds"],
            "mean_return_table_rows": int(main_res["table"].shape[0]),
            "setup_s": {"generate": round(t_gen, 2), "topology+warmup": round(main_res["t_set"], 2)},
            "kernels": {"wrsn_step_kernel_ms": 1e3 * env_launch, "wrsn_obs_kernel_ms": 1e3 * obs_launch,
                        "launch_order_kernels_ms": 1e3 * main_res["order_launch"], "env_steps_per_launch": units},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": per_unit * units, "algorithmic_bytes_per_env_step": {"physics": phys_b, "observation": obs_b},
                         "whole_step_achieved_GBps": (phys_b + obs_b) * value / 1e9, "whole_step_frac": (phys_b + obs_b) * value / 1e9 / peak},
        }
        if rocprof_us:                                          # the same fraction on the committed rocprofv3 duration of that kernel (same configuration)
            out["roofline"]["rocprof_avg_us"] = rocprof_us
            out["roofline"]["frac_rocprof"] = per_unit * units / (rocprof_us * 1e-6) / 1e9 / peak
        zshare = main_res["zero_steps_all"] / max(1.0, env_steps)
        out["value_time_advancing"] = value * (1.0 - zshare)   # completed WRSN.step calls that ran simulated time (the rest: bookkeeping returns at t = warm_up_time)
        if main_res.get("ext"):
            x = main_res["ext"]
            out["extended"] = {"steps": x["steps"], "value": x["env_steps_all"] / x["elapsed_all"], "ms_per_step": 1e3 * x["elapsed_all"] / x["steps"],
                               "value_time_advancing": (x["env_steps_all"] - x["zero_steps_all"]) / x["elapsed_all"],
                               "note": "the timed region of --steps %d launches lasted %.1f ms; the same loop over %d launches, timed the same way" % (args.steps, 1e3 * elapsed, x["steps"])}
        if blocking is not None:
            out["blocking"] = blocking
        if args.cpu_seconds > 0 and world == 1:
            cores = os.cpu_count() or 1
            try:
                usable = len(os.sched_getaffinity(0))           # what this process may run on (a box hands out a share of the host)
            except Exception:
                usable = cores
            threads = max(1, min(usable, 64))
            v, n, dt = cpu_baseline(scenarios[:256], M, args.cpu_seconds, threads)
            out["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": threads, "host_cores": cores, "usable_cores": usable, "kind": "port",
                                   "sample": "%d oracle env-steps (incl. get_state) in %.1f s on %d threads, same synthetic networks and action distribution" % (n, dt, threads)}
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
