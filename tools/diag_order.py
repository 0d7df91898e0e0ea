import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
order = sys.argv[1]
if order == "lib_first":
    from multi_agent_rl_wrsn_amd import _lib
    lib = _lib.load(); print(lib.wrsn_version())
import torch
print("torch cuda", torch.cuda.is_available())
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
try:
    env = VecWRSN([synth_scenario(1, 100, 80)], None, 2)
    print(order, "OK")
except Exception as e:
    print(order, "FAILED", e)
maps = open("/proc/self/maps").read()
libs = sorted(set(l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l or "librocprofiler" in l))
print("\n".join(libs))
