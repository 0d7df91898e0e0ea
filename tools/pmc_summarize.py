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
