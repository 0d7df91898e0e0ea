import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario, RolloutStats
B = 4096
scs = [synth_scenario(e, 200, 200) for e in range(B)]
env = VecWRSN(scs, None, 3, auto_reset=True)
dev = env.device
gen = torch.Generator(device=dev).manual_seed(0)
stats = RolloutStats(B, 3, dev)
r = env.reset()
for _ in range(5):
    r = env.step(r["agent_id"], torch.rand((B, 3), generator=gen, device=dev, dtype=torch.float64))
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode(1)
def timed(name, f):
    t0 = time.perf_counter(); out = f(); dt = time.perf_counter() - t0
    print("%-28s %.3f ms" % (name, dt * 1e3)); return out
for it in range(3):
    print("--- iter", it)
    a = timed("rand", lambda: torch.rand((B, 3), generator=gen, device=dev, dtype=torch.float64))
    r = timed("env.step", lambda: env.step(r["agent_id"], a))
    x = timed("agent_id == 0", lambda: r["agent_id"] == 0)
    y = timed("reward * mask", lambda: r["reward"] * x)
    timed("buf[:,0] += y", lambda: stats.buf[:, 0].__iadd__(y))
    z = timed("terminal.to(f64)", lambda: r["terminal"].to(torch.float64))
    timed("buf[:,3] += z", lambda: stats.buf[:, 3].__iadd__(z))
    timed("buf[:,5] += 1.0", lambda: stats.buf[:, 5].__iadd__(1.0))
    timed("sync", torch.cuda.synchronize)
