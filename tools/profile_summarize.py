"""Summarise the rocprofv3 passes of tools/profile_round.sh into profiles/<tag>_bench_kernel_stats.csv and
profiles/<tag>_traffic.json (run on the GPU box; the outputs come back through gpurun_out/)."""
import collections, csv, glob, json, os, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
# configuration of the profiled bench command (bench.py attaches `traffic` only to a run of the same configuration)
cfg = {"envs_per_gpu": 4096, "nodes": 200, "targets": 200, "chargers": 3, "map_size": 100, "step_budget": 1250}
for a in sys.argv[3:]:
    k, v = a.split("="); cfg[k] = int(v)
min_grid = 64 * cfg["envs_per_gpu"] // 16     # the bench launches only (a step call may come as two launches over half the batch each)
dst = os.path.join(os.path.dirname(out), "profiles_" + tag)
os.makedirs(dst, exist_ok=True)
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, tag + "_bench_kernel_stats.csv"))
def pmc(sub, name):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        if "wrsn_sort_kernel" in r["Kernel_Name"]: agg["_calls"][1] += 1      # one launch-order sort per step call of this pass
        k = "wrsn_step_kernel" if "wrsn_step_kernel" in r["Kernel_Name"] else ("wrsn_obs_kernel" if "wrsn_obs_kernel" in r["Kernel_Name"] else None)
        if k is None or int(r["Grid_Size"]) < min_grid: continue          # the bench launches only (4096 environments)
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    return agg
# average duration per kernel from the --kernel-trace --stats pass (the lean / full instantiations of the step kernel listed separately)
dur = {}
if st:
    for r in csv.DictReader(open(st[0])):
        for k in ("wrsn_step_kernel", "wrsn_obs_kernel", "wrsn_sort_kernel", "wrsn_estimate_kernel"):
            if k in r["Name"]:
                key = k + ("<lean>" if ", false>" in r["Name"] else ("<full>" if ", true>" in r["Name"] else ""))
                dur[key] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
# dispatches of a kernel per step call (pipelined step calls launch the step / observation kernels twice): per-call figures below
calls = max(1, dur.get("wrsn_sort_kernel", {}).get("calls", 0))
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md HBM section) of "
               "`python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --kernel-steps 5 --no-blocking-run` (default step budget); counters are KiB per dispatch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
               "(gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled, an upper bound for narrower accesses)", "config": cfg, "kernels": {}}
for k in ("wrsn_step_kernel", "wrsn_obs_kernel"):
    if fe[k][1] and wr[k][1]:
        per_call = max(1, round(fe[k][1] / max(1, fe["_calls"][1] or calls)))   # 2 with the two-stage pipeline: the figures are per step CALL
        f = fe[k][0] / fe[k][1] * per_call; w = wr[k][0] / wr[k][1] * per_call
        res["kernels"][k] = {"fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024, "dispatches": fe[k][1], "dispatches_per_step_call": per_call}
        cand = [v for kk, v in dur.items() if kk.startswith(k)]
        if cand:                                                # the instantiation with the most calls is the one of the bench loop
            res["kernels"][k]["rocprof_avg_us"] = max(cand, key=lambda v: v["calls"])["avg_us"]     # per dispatch (two overlapping dispatches per call when pipelined)
res["kernel_durations"] = dur
json.dump(res, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
