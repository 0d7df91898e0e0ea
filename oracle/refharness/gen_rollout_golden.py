"""TEST INFRASTRUCTURE ONLY -- golden vector for the roll-out bookkeeping of the reference (controller/ippo/IPPO.py:119-210), produced
by RUNNING the reference's own `roll_out` in this container.

    python oracle/refharness/gen_rollout_golden.py        -> tests/golden_policy/rollout_bookkeeping.npz

`IPPO.roll_out` cannot run against the reference's environment (it logs `request["detailed_rewards"]`, a key `WRSN.step` never
returns, IPPO.py:162) and its module cannot be imported here (tensorboard).  The method is taken as a syntax-tree node out of the
reference's file, compiled unchanged and executed on a stand-in for `self` whose environment is the scripted one of
tests/script_env.py (which does supply that key), whose `get_action` is the scripted policy, and whose `cal_rt_adv` RECORDS what the
method hands it -- the per-episode, per-charger lists of stored states / rewards / next states / terminal flags, i.e. the bookkeeping of
IPPO.py:137-155 -- and returns tagged values.  Stored: those lists in call order, the number of decisions taken, and the batches the
method returns after its reward-outlier selection (IPPO.py:193-209; `np.random.seed` set right before the call)."""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_ROOT = os.environ.get("WRSN_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "tests"))
from script_env import ScriptEnv, scripted_policy  # noqa: E402

N_AGENT, BATCH, ENV_SEED, SELECT_SEED = 3, 10, 77, 5


def main():
    sys.dont_write_bytecode = True
    import torch
    src = open(os.path.join(REF_ROOT, "controller/ippo/IPPO.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "IPPO"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "roll_out"][0]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "IPPO.py:roll_out", "exec"), ns)
    calls = []; n_dec = [0]

    def get_action(agent_id, state):
        a, lp = scripted_policy(n_dec[0], agent_id); n_dec[0] += 1
        return a, lp

    def cal_rt_adv(id, states, rewards, next_states, terminals):
        c = len(calls)
        calls.append((id, states.numpy().copy(), rewards.numpy().copy(), next_states.numpy().copy(), terminals.numpy().copy()))
        n = len(rewards)
        return [torch.tensor(1000.0 * c + q) for q in range(n)], [torch.tensor(2000.0 * c + q) for q in range(n)], [torch.tensor(3000.0 * c + q) for q in range(n)]

    me = types.SimpleNamespace(num_agent=N_AGENT, batch_size=BATCH, device="cpu", env=ScriptEnv(ENV_SEED, N_AGENT), get_action=get_action, cal_rt_adv=cal_rt_adv,
                               writer_log_all=types.SimpleNamespace(writerow=lambda row: None), file_log_all=types.SimpleNamespace(flush=lambda: None),
                               loggers=[{"ep_lens": [], "ep_lifetime": [], "rewards": []} for _ in range(N_AGENT)])
    np.random.seed(SELECT_SEED)
    res = ns["roll_out"](me)
    names = ["states", "actions", "log_probs", "rewards", "next_states", "advantages", "returns", "values"]
    out = {"shape": np.array([N_AGENT, BATCH, ENV_SEED, SELECT_SEED]), "n_decisions": np.array(n_dec[0]), "n_calls": np.array(len(calls)),
           "call_agent": np.array([c[0] for c in calls])}
    for q, c in enumerate(calls):
        out["call%d_states" % q], out["call%d_rewards" % q], out["call%d_next_states" % q], out["call%d_terminals" % q] = c[1], c[2], c[3], c[4]
    for name, per_agent in zip(names, res):
        for a in range(N_AGENT):
            out["batch_%s_%d" % (name, a)] = per_agent[a].numpy()
    os.makedirs(os.path.join(REPO, "tests", "golden_policy"), exist_ok=True)
    np.savez_compressed(os.path.join(REPO, "tests", "golden_policy", "rollout_bookkeeping.npz"), **out)
    print("rollout_bookkeeping.npz: %d decisions, %d episodes x chargers with stored transitions, stored per charger %s" % (
        n_dec[0], len(calls), [int(sum(len(c[2]) for c in calls if c[0] == a)) for a in range(N_AGENT)]))


if __name__ == "__main__":
    main()
