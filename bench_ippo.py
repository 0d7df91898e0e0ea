"""bench_ippo.py -- BASELINE.json configs[2]: 4096 environments x 200 nodes x 3 chargers, IPPO (alg_args/ippo.yaml) roll-out + train
on one MI355X, environment time and policy time reported separately (SURVEY.md 8d "Config 3/4 note": the roll-out is
policy-bound -- the UNet forward is ~5 GFLOP per decision against ~1e-3 GFLOP-equivalents of environment work).

    python bench_ippo.py [--iters 1] [--envs 4096] [--batch-size 512]

Prints ONE JSON line: per training iteration the wall time split into environment launches (VecWRSN.step incl. the
observation), policy inference (UNet forward + sampling for every environment that carries a request), roll-out glue
(transition buffers, density map -> action) and the PPO update; env-steps/s of the roll-out alone and with training.
`python bench.py` stays the headline (random policy) measurement.

`python bench_ippo.py --gpus N` (BASELINE configs[3]: N = 8, 8 x 4096 environments) starts the N ranks itself; under torchrun
(`python -m torch.distributed.run --nproc-per-node N bench_ippo.py --gpus N`) it joins the launcher's group.  Every rank rolls out its own environment shard, the actor / critic gradients are averaged with one RCCL all-reduce per minibatch
(`PPOLearner`), the roll-out returns table is all-gathered once per run, and rank 0 prints the line with whole-job totals."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); without a launcher (WORLD_SIZE unset) this process starts them itself")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--warmup-iters", type=int, default=1, help="untimed iterations first (MIOpen kernel search for every new convolution shape)")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--minibatch-size", type=int, default=64)
    ap.add_argument("--updates", type=int, default=5)
    ap.add_argument("--step-budget", type=int, default=1250)
    ap.add_argument("--infer-chunk", type=int, default=512)
    ap.add_argument("--inference-dtype", default=None, choices=[None, "bf16", "fp16"], help="reduced-precision roll-out inference (update stays float32)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # launcher-free multi-rank entry: before this process touches torch or the GPU
        from multi_agent_rl_wrsn_amd.sharding import launch_ranks
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    import numpy as np
    import torch
    from multi_agent_rl_wrsn_amd import BatchedIPPO, RolloutStats, VecWRSN, init_distributed, synth_scenario
    rank, world, local_rank = init_distributed()
    if world != args.gpus and not (args.gpus == 1 and "WORLD_SIZE" in os.environ):      # torchrun without --gpus: WORLD_SIZE rules
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch.distributed if world > 1 else None
    if dist:                                                   # the RCCL group the ranks really formed
        one = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank)); dist.all_reduce(one)
        if int(one[0]) != world or dist.get_backend() != "nccl":
            raise SystemExit("process group has %d ranks on backend %s, expected %d on nccl" % (int(one[0]), dist.get_backend(), world))
    torch.manual_seed(0); np.random.seed(rank)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    B, N, M = args.envs, args.nodes, 3
    env = VecWRSN([synth_scenario(rank * B + e, N, N) for e in range(B)], None, M, auto_reset=True, step_budget=args.step_budget, device=str(dev),
                  reuse_obs=True)                             # BatchedIPPO only reads the state tensor (index_select / copies)
    algo = BatchedIPPO(dict(batch_size=args.batch_size, minibatch_size=args.minibatch_size, n_updates_per_iteration=args.updates), env,
                       capacity=max(2 * args.batch_size, 4096), infer_chunk=args.infer_chunk, inference_dtype=args.inference_dtype,
                       min_bucket=args.infer_chunk)           # ONE batch shape for every inference pass: MIOpen searches its convolution kernels per shape (seconds each)
    if args.warmup_iters > 0:
        algo.train(args.warmup_iters - 1)
        for k in algo.timers: algo.timers[k] = 0 if isinstance(algo.timers[k], int) else 0.0
    torch.cuda.synchronize(dev)
    c0 = env.counters(); t0 = time.perf_counter()
    rows = algo.train(args.iters - 1)                          # train() runs iterations 0..n inclusive like the reference's loop (IPPO.py:220)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    c1 = env.counters(); t = algo.timers; it = args.iters
    steps = c1["env_steps"] - c0["env_steps"]
    table = RolloutStats.gather_table(env.rollout_table())     # the one exchange of the environment path (RCCL all-gather)
    if dist:
        tot = torch.tensor([steps], dtype=torch.float64, device=dev); dist.all_reduce(tot); steps_all = float(tot[0])
        wl = torch.tensor([wall], dtype=torch.float64, device=dev); dist.all_reduce(wl, op=dist.ReduceOp.MAX); wall = float(wl[0])
    else:
        steps_all = float(steps)
    if rank != 0:
        if dist: dist.destroy_process_group()
        return
    out = {"metric": "IPPO roll-out + train, 4096 envs/GPU x 200 nodes x 3 MC (BASELINE configs[2]; configs[3] under torchrun)", "n_gpus": world,
           "returns_table_rows": int(table.shape[0]), "env_steps_all_ranks": steps_all, "env_steps_per_s_with_training_all_ranks": steps_all / wall, "iterations": it, "warmup_iterations": args.warmup_iters,
           "config": {"workload": "%d envs x %d nodes x %d MC, UNet actor + CNN critic per charger, density-map actions, batch %d / minibatch %d / %d epochs" %
                      (B, N, M, args.batch_size, args.minibatch_size, args.updates), "step_budget": args.step_budget,
                      "policy": "float32, channels-last%s" % ("" if not args.inference_dtype else ", %s roll-out inference" % args.inference_dtype)},
           "per_iteration_s": {"environment": t["env_s"] / it, "policy_inference": t["policy_s"] / it, "rollout_glue": t["glue_s"] / it, "ppo_update": t["train_s"] / it,
                               "wall": wall / it},
           "launches_per_iteration": t["launches"] / it, "requests_served": t["requests"], "env_steps": steps,
           "env_steps_per_s_environment_only": steps / max(t["env_s"], 1e-9), "env_steps_per_s_rollout": steps / max(t["env_s"] + t["policy_s"] + t["glue_s"], 1e-9),
           "env_steps_per_s_with_training": steps / wall, "transitions_per_agent": algo.buffers.counts(),
           "last_rows": rows[-M:], "dtype": "f32 policy / f64 physics", "data": "synthetic"}
    print(json.dumps(out, default=float), flush=True)
    if dist: dist.destroy_process_group()


if __name__ == "__main__":
    main()
