"""Diagnostic (not part of the product path): per-launch time of the environment kernel versus batch size / policy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from multi_agent_rl_wrsn_amd import _lib
if os.environ.get("WRSN_DIAG_LIB"):       # diagnostic override (tools only): an alternative build of the HIP library
    _lib._lib = _lib.bind(C.CDLL(os.environ["WRSN_DIAG_LIB"]))
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario

def run(B, mode, steps=16, N=200, M=3):
    scs = [synth_scenario(e, N, N) for e in range(B)]
    env = VecWRSN(scs, None, M, auto_reset=True, render=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    r = env.reset()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    prev = env.env_info()
    for k in range(steps):
        a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
        if mode == "nocharge": a[:, 2] = 0
        if mode == "shortcharge": a[:, 2] *= 0.05
        ev0.record(); r = env.step(r["agent_id"], a); ev1.record(); torch.cuda.synchronize()
        info = env.env_info(); st = r["status"].cpu().numpy(); real = st != 3
        dt = (info["n_ticks"] - prev["n_ticks"])[real]; dx = (info["n_exact"] - prev["n_exact"])[real]
        rows.append((ev0.elapsed_time(ev1), real.sum(), dt.mean() if real.any() else 0, dt.max() if real.any() else 0, dx.sum()))
        prev = info
    ms = np.array([x[0] for x in rows[4:]])
    print("B=%5d mode=%-11s ms/launch mean %.2f max %.2f | steps/launch %.0f ticks mean %.0f max %.0f exact/launch %.0f" % (
        B, mode, ms.mean(), ms.max(), np.mean([x[1] for x in rows[4:]]), np.mean([x[2] for x in rows[4:]]), np.max([x[3] for x in rows[4:]]), np.mean([x[4] for x in rows[4:]])), flush=True)
    for x in rows[:10]: print("    %.2f ms  real %d  ticks mean %.0f max %.0f exact %d" % x)
    env.close()

if __name__ == "__main__":
    if os.environ.get("WRSN_DIAG_QUICK"):
        run(256, "random"); run(4096, "random"); run(4096, "nocharge")
    else:
        for B in (256, 1024, 4096):
            run(B, "random")
        run(4096, "nocharge")
        run(4096, "shortcharge")
