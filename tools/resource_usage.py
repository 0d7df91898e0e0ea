"""Compile the product library with -Rpass-analysis=kernel-resource-usage and write profiles/<tag>_kernel_resource_usage.csv
(registers, spills, scratch and occupancy of every kernel).  CPU only: hipcc cross-compiles gfx950.  python tools/resource_usage.py r02"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
csrc = os.path.join(ROOT, "multi_agent_rl_wrsn_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage",
                        "-o", os.path.join(td, "x.so"), "wrsn_api.hip"], cwd=csrc, stderr=subprocess.PIPE, text=True)
rows, cur = [], None
keys = [("VGPRs", "VGPRs"), ("AGPRs", "AGPRs"), ("TotalSGPRs", "SGPRs"), ("SGPRs Spill", "SGPR_spill"), ("VGPRs Spill", "VGPR_spill"),
        ("ScratchSize [bytes/lane]", "scratch_bytes_per_lane"), ("Occupancy [waves/SIMD]", "occupancy_waves_per_simd")]
for line in p.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"kernel": m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for k, name in keys:
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
        if m: cur[name] = m.group(1)
out = os.path.join(ROOT, "profiles", tag + "_kernel_resource_usage.csv")
with open(out, "w") as w:
    w.write("kernel," + ",".join(n for _, n in keys) + "\n")
    for r in rows:
        if "VGPRs" in r: w.write(r["kernel"] + "," + ",".join(r.get(n, "") for _, n in keys) + "\n")
print("wrote", out, len(rows), "kernels")
