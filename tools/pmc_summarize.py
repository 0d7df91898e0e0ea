"""Summarise the counter passes of tools/pmc_sq.sh: per-kernel mean of every counter over the dispatches of the workload."""
import collections, csv, glob, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "wrsn" not in k or int(r["Grid_Size"]) < 64 * 4096: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(os.path.join(out, "pmc_summary.csv"), "w") as w:
    w.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v, n = agg[k][c]
            w.write("%s,%s,%.1f,%d\n" % (k, c, v / n, n)); print("%-42s %-28s %16.1f  (%d)" % (k, c, v / n, n))
        m = {c: agg[k][c][0] / agg[k][c][1] for c in agg[k]}
        if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_THREAD_CYCLES_VALU"):
            # thread-cycles / instruction-cycles of the VALU = lanes that were enabled, on average, while a VALU instruction executed (of 64)
            lanes = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
            w.write("%s,derived:active_lanes_per_VALU_instruction,%.2f,%d\n" % (k, lanes, agg[k]["SQ_THREAD_CYCLES_VALU"][1])); print("%-42s %-28s %16.2f" % (k, "active lanes / VALU instr", lanes))
        if m.get("SQ_WAVE_CYCLES") and m.get("SQ_ACTIVE_INST_VALU"):
            w.write("%s,derived:VALU_share_of_wave_cycles,%.4f,%d\n" % (k, m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], agg[k]["SQ_WAVE_CYCLES"][1]))
