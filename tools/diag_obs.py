import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_agent_rl_wrsn_amd import _lib
if os.environ.get("WRSN_DIAG_LIB"): _lib._lib = _lib.bind(C.CDLL(os.environ["WRSN_DIAG_LIB"]))
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = 4096
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, reuse_obs=True)
r = env.reset(); g = torch.Generator(device="cuda").manual_seed(0)
for _ in range(4): r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
ids = r["agent_id"].clamp(min=0).to(torch.int32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); ts = []
for _ in range(10):
    e0.record(); env.render_state(ids, out=env.state); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(os.environ.get("WRSN_DIAG_LIB", "product"), "obs ms: %.3f" % (sum(ts[2:]) / len(ts[2:])))
