"""TEST INFRASTRUCTURE ONLY -- golden vector for the PPO update of the reference (controller/ippo/IPPO.py:229-271), produced by
RUNNING the reference's own statements in this container.

    python oracle/refharness/gen_update_golden.py        -> tests/golden_policy/update_step.npz

The update is not a function of the reference: it is the loop `for _ in range(self.n_updates_per_iteration):` inside
`IPPO.train` (IPPO.py:230), behind the roll-out, TensorBoard writers and CSV logging of that method (tensorboard is not installed
here, so the module cannot even be imported).  This script takes that loop -- and the methods `evaluate` / `get_value` it calls -- as
syntax-tree nodes out of the reference's source file, compiles the nodes unchanged and executes them on a stand-in for `self`
that carries the reference's own networks (`UNet`, `CNNCritic`, imported from the reference), an Adam optimiser built as
IPPO.py:65-69 builds it, and the hyper-parameters of alg_args/ippo.yaml at a small batch (16 rows, minibatches of 8, 2 epochs).
Nothing of the reference's text is stored: inputs are closed formulas of the element index (the test regenerates them), weights
come from `formula_fill`, and the file holds the losses the loop logged plus, for every parameter / buffer after the update, its
sum, absolute sum and first four values."""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_ROOT = os.environ.get("WRSN_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, HERE)
from gen_policy_golden import formula_fill  # noqa: E402

G, SHUFFLE_SEED = 100, 123
# two cases: `split` -- 16 rows in minibatches of 8, 2 epochs: shuffling, slicing and the regime a roll-out really leaves behind (the stored
# log-probabilities belong to another BatchNorm batch, every ratio is clipped); `whole` -- one minibatch that IS the batch the stored
# log-probabilities were taken over: ratios around 1, the unclipped policy gradient
CASES = {"split": (16, 8, 2), "whole": (8, 8, 2)}


def formula_batch(torch, B):
    """The batch the update runs on, as closed formulas (tests/test_ippo.py builds the same tensors)."""
    n = B * 4 * G * G
    i = torch.arange(n, dtype=torch.float64)
    states = (0.5 + 0.5 * torch.sin(0.0131 * i + 0.7)).reshape(B, 4, G, G)
    states[:, 0] *= 3.0                                        # map 1 of an observation reaches several units
    j = torch.arange(B * G * G, dtype=torch.float64)
    actions = (0.3 * torch.sin(0.0173 * j + 0.2)).reshape(B, G, G)
    k = torch.arange(B, dtype=torch.float64)
    return dict(states=states.float(), actions=actions.float(),
                log_probs=(-9000.0 + 40.0 * torch.sin(0.9 * k)).float(), advantages=torch.cos(1.3 * k + 0.1).float(),
                returns=(0.5 * torch.sin(0.7 * k) + 0.2).float(), values=(0.4 * torch.sin(0.7 * k + 0.3)).float())


def reference_nodes():
    src = open(os.path.join(REF_ROOT, "controller/ippo/IPPO.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "IPPO"][0]
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    loops = [n for n in ast.walk(fns["train"]) if isinstance(n, ast.For) and isinstance(n.iter, ast.Call)
             and ast.unparse(n.iter) == "range(self.n_updates_per_iteration)"]
    assert len(loops) == 1, "IPPO.train: expected one update loop"
    return fns["evaluate"], fns["get_value"], loops[0]


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_ROOT)
    os.chdir(REF_ROOT)
    import torch
    from torch import nn
    from torch.distributions import Normal
    from controller.ppo.actor.UnetActor import UNet
    from controller.ppo.critic.CNNCritic import CNNCritic
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ev_node, gv_node, loop_node = reference_nodes()
    ns = {"torch": torch, "np": np, "nn": nn, "Normal": Normal}
    exec(compile(ast.Module(body=[ev_node, gv_node], type_ignores=[]), "IPPO.py:evaluate/get_value", "exec"), ns)
    out = {"shape": np.array([G, SHUFFLE_SEED]), "cases": np.array(sorted(CASES))}
    for case in sorted(CASES):
        run_case(torch, nn, ns, loop_node, UNet, CNNCritic, case, out)
    os.makedirs(os.path.join(REPO, "tests", "golden_policy"), exist_ok=True)
    np.savez_compressed(os.path.join(REPO, "tests", "golden_policy", "update_step.npz"), **out)


def run_case(torch, nn, ns, loop_node, UNet, CNNCritic, case, out):
    B, MB, EPOCHS = CASES[case]
    actor, critic = UNet(), CNNCritic()
    formula_fill(actor); formula_fill(critic)
    actor.train(); critic.train()                              # IPPO.py never calls .eval()
    me = types.SimpleNamespace(actors=[actor], critics=[critic], loggers=[{"losses": []}],
                               optimizers=[torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=3.0e-4)],   # IPPO.py:65-69, ippo.yaml
                               clip=0.2, norm_adv=True, clip_vloss=True, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5,
                               batch_size=B, minibatch_size=MB, n_updates_per_iteration=EPOCHS)
    me.evaluate = types.MethodType(ns["evaluate"], me); me.get_value = types.MethodType(ns["get_value"], me)
    b = formula_batch(torch, B)
    # stored log-probabilities near what the actor gives (ratios around 1, both clip branches taken): the reference's `evaluate` on a COPY
    # of the actor (a training-mode forward moves the BatchNorm running statistics) over the whole batch, plus a spread; stored in the file
    import copy
    probe = types.SimpleNamespace(actors=[copy.deepcopy(actor)])
    with torch.no_grad():
        lp0, _ = types.MethodType(ns["evaluate"], probe)(0, b["states"], b["actions"])
    b["log_probs"] = (lp0.double() + 0.15 * torch.sin(1.7 * torch.arange(B, dtype=torch.float64))).float()
    env = {"self": me, "id": 0, "torch": torch, "np": np, "nn": nn, "b_inds": np.arange(B), "clipfracs": [],
           "batch_states": b["states"], "batch_actions": b["actions"], "batch_log_probs": b["log_probs"],
           "batch_advantages": b["advantages"], "batch_returns": b["returns"], "batch_values": b["values"]}
    np.random.seed(SHUFFLE_SEED)
    exec(compile(ast.Module(body=[loop_node], type_ignores=[]), "IPPO.py:update-loop", "exec"), env)
    out[case + "_losses"] = np.array([float(v) for v in me.loggers[0]["losses"]]); out[case + "_clipfracs"] = np.array(env["clipfracs"])
    out[case + "_shape"] = np.array([B, MB, EPOCHS]); out[case + "_log_probs"] = b["log_probs"].numpy()
    for tag, net in (("actor", actor), ("critic", critic)):
        names, stats = [], []
        for name, t in net.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            f = t.detach().double().flatten()
            names.append(name); stats.append([float(f.sum()), float(f.abs().sum())] + [float(v) for v in f[:4]] + [0.0] * max(0, 4 - f.numel()))
        out[case + "_" + tag + "_names"] = np.array(names); out[case + "_" + tag + "_stats"] = np.array(stats)
    print("update_step.npz[%s]: %d minibatch steps, losses %s, clipfracs %s" % (case, len(out[case + "_losses"]), np.round(out[case + "_losses"], 5), out[case + "_clipfracs"]))


if __name__ == "__main__":
    main()
