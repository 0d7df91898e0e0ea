"""Diagnostic: phase durations inside wrsn_obs_kernel (needs tools/libwrsn_obs_prof.so, -DWRSN_OBS_PROF -DWRSN_PROFILE)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwrsn_obs_prof.so")))
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
B = 4096
env = VecWRSN([synth_scenario(e, 200, 200) for e in range(B)], None, 3, auto_reset=True, reuse_obs=True)
r = env.reset(); g = torch.Generator(device="cuda").manual_seed(0)
for _ in range(4): r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
ids = r["agent_id"].clamp(min=0).to(torch.int32)
if len(sys.argv) > 1 and sys.argv[1] == "full": env._h.set_obs_reuse(False)      # every render draws map 1 (default: map 1 of the row is reused)
for _ in range(3): env.render_state(ids, out=env.state)
torch.cuda.synchronize()
a = np.zeros((B * 25,), dtype=np.int64)
_lib.check(env._h.lib, env._h.lib.wrsn_peek(env._h._h, 10, a.ctypes.data))
p = a[:B * 24].reshape(B, 24)
names = ["prologue+fill0", "maps234 (even blocks)", "main loop", "map1 store", "maps234 (odd blocks)"]
for q, n in enumerate(names):
    print("%-24s mean %9.0f  even-blocks %9.0f  odd-blocks %9.0f" % (n, p[:, q].mean(), p[0::2, q].mean(), p[1::2, q].mean()))
t0 = p[:, 5]; tot = p[:, :5].sum(1)
print("block duration mean %.0f cycles; first start %d last end %d span %d" % (tot.mean(), t0.min(), (t0 + tot).max(), (t0 + tot).max() - t0.min()))
