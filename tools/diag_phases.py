"""Diagnostic: per-phase cycle totals of the environment kernel (needs tools/libwrsn_hip_profile.so)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from multi_agent_rl_wrsn_amd import _lib
_lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "tools", "libwrsn_hip_profile.so")))   # diagnostic override, tools only
from multi_agent_rl_wrsn_amd import VecWRSN, synth_scenario
NAMES = ["scalar_run", "grid_run", "steady_batch", "update_reward", "exact_walk", "rebuild_cache", "set_levels", "min_fitness",
         "generic_items(incl)", "ur_flush", "load", "store", "#services", "#fused_s", "#jumped_s", "#generic_items",
         "xw:receivers", "xw:single", "#single", "grid:per_second_loop", "#per_second_s", "#locate", "grid:loop_head", "#grid_iterations"]
def prof(env):
    a = np.zeros((env.num_env * 25,), dtype=np.int64)
    _lib.check(env._h.lib, env._h.lib.wrsn_peek(env._h._h, 10, a.ctypes.data))
    return a[:env.num_env * 24].reshape(env.num_env, 24).copy(), a[env.num_env * 24:].copy()
if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    NN = int(os.environ.get("WRSN_N", "200")); MM = int(os.environ.get("WRSN_M", "3"))
    scs = [synth_scenario(e, NN, NN) for e in range(B)]
    env = VecWRSN(scs, None, MM, auto_reset=True, render=False, step_budget=int(os.environ.get("WRSN_BUDGET", "0")))
    g = torch.Generator(device="cuda").manual_seed(1)
    r = env.reset()
    for k in range(4):
        r = env.step(r["agent_id"], torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64))
    torch.cuda.synchronize()
    p0, k0 = prof(env)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for k in range(10):
        a = torch.rand((B, 3), generator=g, device="cuda", dtype=torch.float64)
        e0.record(); r = env.step(r["agent_id"], a); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    p1, k1 = prof(env)
    d = p1 - p0; kd = k1 - k0
    print("B=%d launches=10 ms/launch %s" % (B, np.round(ms, 2)))
    print("kernel cycles per env: mean %.0f max %.0f (100 MHz realtime? clock64 = shader clock)" % (kd.mean(), kd.max()))
    worst = np.argsort(-kd)[:5]
    print("%-16s %14s %14s   worst envs: %s" % ("phase", "mean/env", "max/env", worst))
    for i, n in enumerate(NAMES):
        print("%-16s %14.0f %14.0f   %s" % (n, d[:, i].mean(), d[:, i].max(), d[worst, i]))
