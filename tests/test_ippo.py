"""f2 / f4 on the CPU: policy networks and trainer arithmetic against vectors generated from the reference's own modules
(tests/golden_policy/policy_nets.npz, script oracle/refharness/gen_policy_golden.py), the PPO update against the reference's own update
loop run on its own networks (tests/golden_policy/update_step.npz, oracle/refharness/gen_update_golden.py), the batch selection, the device-side
transition bookkeeping (emulated kernels) against the reference's list bookkeeping restated from controller/ippo/IPPO.py:137-155,
and the data-parallel gradient exchange on a world-2 gloo group."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden_policy", "policy_nets.npz")


def formula_fill(module):
    """the fill of oracle/refharness/gen_policy_golden.py"""
    import torch
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.dtype.is_floating_point:
                continue
            k = (zlib.crc32(name.encode()) % 1000) / 100.0
            v = 0.05 * torch.sin(0.37 * torch.arange(t.numel(), dtype=torch.float64) + k)
            if name.endswith("bn.weight") or name.endswith("running_var"):
                v = v + 1.0
            t.copy_(v.reshape(t.shape).to(t.dtype))


def test_networks_match_the_reference_modules():
    import torch
    from multi_agent_rl_wrsn_amd import build_networks
    z = np.load(GOLD)
    torch.manual_seed(0); torch.set_num_threads(4)
    UNet, CNNCritic = build_networks(100)
    actor, critic = UNet(), CNNCritic()
    # same state_dict keys and shapes: reference checkpoints (actor.pth / critic.pth, IPPO.py:296-309) load unchanged
    assert [n for n in actor.state_dict()] == list(z["actor_names"]) and [str(tuple(v.shape)) for v in actor.state_dict().values()] == list(z["actor_shapes"])
    assert [n for n in critic.state_dict()] == list(z["critic_names"]) and [str(tuple(v.shape)) for v in critic.state_dict().values()] == list(z["critic_shapes"])
    formula_fill(actor); formula_fill(critic)
    x = torch.from_numpy(z["x"])
    actor.train()
    with torch.no_grad():
        mean, log_std = actor(x)
        value = critic(x)
    assert np.allclose(mean.numpy(), z["actor_mean"], rtol=1e-4, atol=1e-5)
    assert np.allclose(log_std.detach().numpy(), z["actor_log_std"], rtol=0, atol=1e-7)
    assert np.allclose(value.numpy(), z["critic_value"], rtol=1e-4, atol=1e-6)


def _update_batch(torch, B, G):
    """the batch of oracle/refharness/gen_update_golden.py (closed formulas of the element index)"""
    i = torch.arange(B * 4 * G * G, dtype=torch.float64)
    states = (0.5 + 0.5 * torch.sin(0.0131 * i + 0.7)).reshape(B, 4, G, G)
    states[:, 0] *= 3.0
    j = torch.arange(B * G * G, dtype=torch.float64)
    k = torch.arange(B, dtype=torch.float64)
    return dict(states=states.float(), actions=(0.3 * torch.sin(0.0173 * j + 0.2)).reshape(B, G, G).float(), advantages=torch.cos(1.3 * k + 0.1).float(),
                returns=(0.5 * torch.sin(0.7 * k) + 0.2).float(), values=(0.4 * torch.sin(0.7 * k + 0.3)).float())


@pytest.mark.parametrize("case", ["split", "whole"])
def test_update_matches_the_reference_update_loop(case):
    """f4 pinned (r03): `PPOLearner.update` against what the reference's OWN update loop (the `for` node of IPPO.train, IPPO.py:230-271,
    with its `evaluate` / `get_value`, executed by oracle/refharness/gen_update_golden.py on the reference's networks) leaves behind:
    the loss of every minibatch step, the clip fractions, and every parameter and BatchNorm buffer after the update.  `split`: 16 rows
    in shuffled minibatches of 8, every ratio clipped (what a roll-out really leaves); `whole`: the unclipped policy gradient."""
    import torch
    from multi_agent_rl_wrsn_amd import PPOLearner
    z = np.load(os.path.join(ROOT, "tests", "golden_policy", "update_step.npz"))
    G, seed = (int(v) for v in z["shape"]); B, MB, EPOCHS = (int(v) for v in z[case + "_shape"])
    torch.manual_seed(0); torch.set_num_threads(4)
    lr = PPOLearner({"lr": 3.0e-4, "clip": 0.2, "batch_size": B, "minibatch_size": MB, "n_updates_per_iteration": EPOCHS, "norm_adv": True,
                     "clip_vloss": True, "ent_coef": 0.0, "vf_coef": 0.5, "max_grad_norm": 0.5}, 1, G, "cpu")
    formula_fill(lr.actors[0]); formula_fill(lr.critics[0])
    batch = _update_batch(torch, B, G); batch["log_probs"] = torch.from_numpy(z[case + "_log_probs"])
    np.random.seed(seed)
    stats = lr.update(0, batch)
    losses = np.array(lr.loggers[0]["losses"])
    assert losses.shape == z[case + "_losses"].shape and np.allclose(losses, z[case + "_losses"], rtol=2e-4, atol=1e-6), (losses, z[case + "_losses"])
    assert abs(stats[4] - float(np.mean(z[case + "_clipfracs"]))) < 1e-6
    for tag, net in (("actor", lr.actors[0]), ("critic", lr.critics[0])):
        sd = {n: t for n, t in net.state_dict().items() if t.dtype.is_floating_point}
        assert list(sd) == list(z[case + "_" + tag + "_names"])
        for (name, t), ref in zip(sd.items(), z[case + "_" + tag + "_stats"]):
            f = t.detach().double().flatten()
            got = np.array([float(f.sum()), float(f.abs().sum())] + [float(v) for v in f[:4]] + [0.0] * max(0, 4 - f.numel()))
            tol = 2e-5 * max(1.0, float(f.abs().sum())) / max(1, f.numel()) ** 0.5 + 1e-6
            assert np.allclose(got[:2], ref[:2], rtol=0, atol=tol * max(1, f.numel()) ** 0.5), (tag, name, got, ref)
            assert np.allclose(got[2:], ref[2:], rtol=1e-4, atol=2e-6), (tag, name, got, ref)


def test_cal_rt_adv_matches_the_reference_function():
    import torch
    from multi_agent_rl_wrsn_amd import PPOLearner
    z = np.load(GOLD)
    T = len(z["rt_rewards"])
    table = {float(i): float(v) for i, v in enumerate(z["rt_values"])}; table.update({float(i) + 100: float(v) for i, v in enumerate(z["rt_next_values"])})
    lr = PPOLearner(dict(gamma=0.99, gae=True, gae_lambda=0.95), 1, 12, "cpu")
    lr._values = lambda id, s: torch.tensor([table[float(v)] for v in s])
    states = torch.arange(T, dtype=torch.float32)
    ret, adv, v = lr.cal_rt_adv(0, states, torch.from_numpy(z["rt_rewards"]), states + 100, torch.from_numpy(z["rt_terminals"]))
    assert np.allclose(ret.numpy(), z["rt_returns_gae1"], rtol=1e-6, atol=1e-7) and np.allclose(adv.numpy(), z["rt_adv_gae1"], rtol=1e-6, atol=1e-7)
    # with the flags the roll-out really stores (all False, IPPO.py:144-155) returns are the rewards
    ret0, adv0, _ = lr.cal_rt_adv(0, states, torch.from_numpy(z["rt_rewards"]), states + 100, torch.zeros(T))
    assert np.allclose(ret0.numpy(), z["rt_rewards"], atol=1e-7) and np.allclose(adv0.numpy(), z["rt_rewards"] - z["rt_values"], atol=1e-7)
    # the plain (gae: False) branch of the reference raises on every non-empty input; so does this one
    assert str(z["rt_error_gae0"]) == "IndexError"
    lr.gae = False
    with pytest.raises(IndexError):
        lr.cal_rt_adv(0, states, torch.from_numpy(z["rt_rewards"]), states + 100, torch.zeros(T))


def test_select_batch_is_the_reference_selection():
    from multi_agent_rl_wrsn_amd import select_batch
    rng = np.random.RandomState(3)
    rewards = rng.randn(700)
    np.random.seed(11)
    got = select_batch(rewards, 512)
    # IPPO.py:193-200, statement for statement
    np.random.seed(11)
    mean = np.mean(rewards)
    abs_diff = np.abs(rewards - mean)
    indices = np.argsort(abs_diff)
    selected_num = int(512 / 2.0)
    random_num = 512 - selected_num
    want = np.concatenate((indices[-selected_num:], np.random.choice(len(rewards) - selected_num, size=random_num, replace=False)))
    assert np.array_equal(got, want) and len(got) == 512


def _policy(e, n):
    """deterministic stand-in for the actor: action 3-vector and log-prob of environment e's n-th decision"""
    r = np.random.RandomState(1000 * e + n)
    return r.rand(3).astype(np.float32), np.float32(-0.01 * n - e)


def reference_bookkeeping(step, reset, n_agent, n_decisions):
    """controller/ippo/IPPO.py:137-155 for ONE environment: returns per agent the list of
    (prev_state, input_action, log_prob, reward, state, now).  `reset()` / `step(agent, action)` return request dicts."""
    out = [[] for _ in range(n_agent)]
    done = 0
    while done < n_decisions:
        request = reset()
        pre = [None] * n_agent; prev_state = [None] * n_agent; prev_action = [None] * n_agent
        while done < n_decisions:
            a = request["agent_id"]
            action, log_prob = request["policy"](done); done += 1
            pre[a] = log_prob; prev_state[a] = request["state"]; prev_action[a] = action
            request = step(a, action)
            if request["terminal"]:
                break
            b = request["agent_id"]
            if pre[b] is None:
                continue
            out[b].append((prev_state[b], prev_action[b], pre[b], request["reward"], request["state"], request["now"]))
    return out


def test_bookkeeping_restatement_and_selection_match_the_reference_roll_out():
    """f2 pinned (r03): the reference's OWN `roll_out` (IPPO.py:119-210, run by oracle/refharness/gen_rollout_golden.py on the scripted
    environment of tests/script_env.py) against `reference_bookkeeping` above -- the restatement the device-side transition buffers are
    held to -- and against `select_batch`: the stored (state, reward, next state) lists the reference hands `cal_rt_adv` per episode and
    charger, and the batches it returns after its reward-outlier selection (states, actions, log-probabilities, rewards, next states, and
    the positions its returns / advantages / values were taken from)."""
    from multi_agent_rl_wrsn_amd import select_batch
    from script_env import ScriptEnv, scripted_policy
    z = np.load(os.path.join(ROOT, "tests", "golden_policy", "rollout_bookkeeping.npz"))
    n_agent, batch, env_seed, sel_seed = (int(v) for v in z["shape"]); n_dec = int(z["n_decisions"])
    env = ScriptEnv(env_seed, n_agent)

    def wrap(r):
        r = dict(r); r["policy"] = lambda n, a=r["agent_id"]: scripted_policy(n, a); return r
    per_agent = reference_bookkeeping(lambda a, action: wrap(env.step(a, action)), lambda: wrap(env.reset()), n_agent, n_dec)
    # the lists cal_rt_adv received, call by call (one call per episode and charger with stored transitions), and a tag per stored transition
    want = [dict(states=[], rewards=[], next_states=[], tags=[]) for _ in range(n_agent)]
    for c in range(int(z["n_calls"])):
        a = int(z["call_agent"][c]); w = want[a]
        w["states"] += list(z["call%d_states" % c]); w["rewards"] += list(z["call%d_rewards" % c]); w["next_states"] += list(z["call%d_next_states" % c])
        w["tags"] += [(c, q) for q in range(len(z["call%d_rewards" % c]))]
        assert not z["call%d_terminals" % c].any()              # stored transitions are never terminal (IPPO.py:144-145)
    np.random.seed(sel_seed)
    for a in range(n_agent):
        got = per_agent[a]; w = want[a]
        assert len(got) == len(w["rewards"]) >= batch
        assert np.array_equal(np.array([t[0] for t in got]), np.array(w["states"])) and np.array_equal(np.array([t[4] for t in got]), np.array(w["next_states"]))
        assert np.array_equal(np.array([t[3] for t in got], dtype=np.float32), np.array(w["rewards"], dtype=np.float32))
        idx = select_batch([t[3] for t in got], batch)          # the selections of the chargers draw from one stream, in charger order
        assert np.array_equal(z["batch_states_%d" % a], np.array([got[i][0] for i in idx])) and np.array_equal(z["batch_next_states_%d" % a], np.array([got[i][4] for i in idx]))
        assert np.array_equal(z["batch_actions_%d" % a], np.array([got[i][1] for i in idx])) and np.array_equal(z["batch_log_probs_%d" % a], np.array([got[i][2] for i in idx], dtype=np.float32))
        assert np.array_equal(z["batch_rewards_%d" % a], np.array([got[i][3] for i in idx], dtype=np.float32))
        for name, k in (("returns", 1000.0), ("advantages", 2000.0), ("values", 3000.0)):      # the tagged values the stand-in cal_rt_adv returned
            assert np.array_equal(z["batch_%s_%d" % (name, a)], np.array([k * w["tags"][i][0] + w["tags"][i][1] for i in idx], dtype=np.float32)), (name, a)


def test_emulated_transition_buffers_equal_the_reference_bookkeeping():
    """wrsn_rollout_record / wrsn_rollout_collect (csrc/wrsn_rollout.h, emulated) over a batch with auto-reset and a step
    budget == the reference's per-environment list bookkeeping on single environments."""
    import ctypes as C
    from emu_env import EmuVec
    from multi_agent_rl_wrsn_amd import DEFAULT_MC_SPEC, _lib, synth_scenario
    B, M, G, CAP, K = 3, 2, 12, 64, 28
    scs = [synth_scenario(300 + e, 70, 60) for e in range(B)]
    ev = EmuVec(scs, DEFAULT_MC_SPEC, M, map_size=G)
    ev.h.set_step_budget(60)
    S = 4 * G * G
    arrs = dict(pend_state=np.zeros((B, M, S), np.float32), pend_action=np.zeros((B, M, 3), np.float32), pend_logp=np.zeros((B, M), np.float32),
                pend_valid=np.zeros((B, M), np.uint8), state=np.zeros((M, CAP, S), np.float32), action=np.zeros((M, CAP, 3), np.float32),
                next_state=np.zeros((M, CAP, S), np.float32), reward=np.zeros((M, CAP), np.float32), logp=np.zeros((M, CAP), np.float32),
                now=np.zeros((M, CAP), np.float64), env=np.zeros((M, CAP), np.int32), count=np.zeros(M, np.int32))
    buf = _lib.WrsnTransitionBuffers(CAP, 3, *[arrs[k].ctypes.data for k in ("pend_state", "pend_action", "pend_logp", "pend_valid", "state", "action",
                                                                            "next_state", "reward", "logp", "now", "env", "count")])
    ev.reset()
    n_dec = np.zeros(B, dtype=int)
    for it in range(400):
        ids = ev.agent_id.copy()
        act = np.zeros((B, 3), np.float32); lp = np.zeros(B, np.float32)
        for e in range(B):
            if ids[e] >= 0 and n_dec[e] < K:
                act[e], lp[e] = _policy(e, n_dec[e]); n_dec[e] += 1
            elif ids[e] >= 0:
                ids[e] = -2                                    # this environment has had its K decisions
        ev.h.rollout_record(buf, ids.ctypes.data, act.ctypes.data, lp.ctypes.data, ev.obs.ctypes.data)
        ev.step(ids, act.astype(np.float64), auto_reset=True)
        ev.h.rollout_collect(buf, **ev._ptrs(True))
        if (n_dec >= K).all() and not (ev.status == 4).any():
            break
    # the same decisions, one environment at a time, with the reference's bookkeeping
    want = [[] for _ in range(M)]
    for e in range(B):
        one = EmuVec([scs[e]], DEFAULT_MC_SPEC, M, map_size=G)
        def req():
            return dict(agent_id=int(one.agent_id[0]), state=one.obs[0].reshape(-1).copy(), reward=float(one.reward[0]), terminal=bool(one.terminal[0]),
                        now=float(one.now[0]), policy=lambda n, e=e: _policy(e, n))
        def reset():
            one.reset(); return req()
        def step(a, action):
            one.step([a], np.asarray(action, np.float64)[None]); return req()
        per_agent = reference_bookkeeping(step, reset, M, K)
        for a in range(M):
            want[a] += [(e,) + t for t in per_agent[a]]
    for a in range(M):
        n = int(arrs["count"][a])
        assert n == len(want[a]) and 0 < n <= CAP
        got = sorted(range(n), key=lambda q: (arrs["env"][a, q], arrs["now"][a, q], arrs["logp"][a, q]))
        ref = sorted(want[a], key=lambda t: (t[0], t[6], t[3]))
        for q, t in zip(got, ref):
            assert arrs["env"][a, q] == t[0] and arrs["now"][a, q] == t[6]
            assert np.array_equal(arrs["state"][a, q], t[1]) and np.array_equal(arrs["action"][a, q], t[2]) and arrs["logp"][a, q] == t[3]
            assert arrs["reward"][a, q] == np.float32(t[4]) and np.array_equal(arrs["next_state"][a, q], t[5])


def _flat_params(lr):
    import torch
    return torch.cat([p.detach().reshape(-1) for p in list(lr.actors[0].parameters()) + list(lr.critics[0].parameters())]).clone()


def _dp_batch(seed, n, G):
    import torch
    g = torch.Generator().manual_seed(seed)
    return dict(states=torch.rand((n, 4, G, G), generator=g), actions=torch.randn((n, G, G), generator=g), log_probs=torch.randn(n, generator=g) - 150.0,
                advantages=torch.randn(n, generator=g), returns=torch.randn(n, generator=g), values=torch.randn(n, generator=g))


_DP_ARGS = dict(batch_size=8, minibatch_size=4, n_updates_per_iteration=2, lr=1e-3)
_DP_G = 12


def _dp_worker(rank, world, port, q, same_batch=False):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_agent_rl_wrsn_amd import PPOLearner
    torch.manual_seed(100 + rank)                              # different initial weights per rank: the broadcast must equalise them
    lr = PPOLearner(_DP_ARGS, 1, _DP_G, "cpu")
    if rank == 0:
        formula_fill(lr.actors[0]); formula_fill(lr.critics[0])
        for net in (lr.actors[0], lr.critics[0]):              # what the constructor did, once more with the closed-formula weights
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t.data, src=0)
    else:
        for net in (lr.actors[0], lr.critics[0]):
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t.data, src=0)
    p0 = _flat_params(lr)
    batch = _dp_batch(7 + (0 if same_batch else rank), _DP_ARGS["batch_size"], _DP_G)   # a different local batch per rank (or the same one)
    order = np.random.RandomState(3)
    lr.update(0, batch, shuffle=order.shuffle)
    q.put((rank, p0.numpy(), _flat_params(lr).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(same_batch):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, same_batch)) for r in range(2)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in ps)
    return res


def _single_process_reference(batches):
    """What the data-parallel update has to equal: ONE process, the same initial weights, and per minibatch the gradient that is the
    mean of the ranks' gradients -- computed here by hand from one backward pass per rank batch (IPPO.py:225-271 otherwise unchanged:
    the two clip_grad_norm_ calls and the Adam step act on the averaged gradient)."""
    import torch
    from multi_agent_rl_wrsn_amd import PPOLearner
    torch.set_num_threads(2)
    lr = PPOLearner(_DP_ARGS, 1, _DP_G, "cpu")
    formula_fill(lr.actors[0]); formula_fill(lr.critics[0])
    p0 = _flat_params(lr)
    params = list(lr.actors[0].parameters()) + list(lr.critics[0].parameters())
    order = np.random.RandomState(3)
    b_inds = np.arange(lr.batch_size)
    for _ in range(lr.n_updates_per_iteration):
        order.shuffle(b_inds)
        for start in range(0, lr.batch_size, lr.minibatch_size):
            mb = torch.as_tensor(b_inds[start:start + lr.minibatch_size], dtype=torch.long)
            grads = []
            for batch in batches:
                lr.optimizers[0].zero_grad()
                lr.minibatch_loss(0, batch, mb)[0].backward()
                grads.append([p.grad.detach().clone() for p in params])
            for k, p in enumerate(params):
                acc = grads[0][k].clone()
                for g in grads[1:]:
                    acc += g[k]                                # the all-reduce sums ...
                p.grad.copy_(acc / len(grads))                 # ... and the bucket is divided by the world size
            lr.world = 1                                       # (no process group here: apply_gradients must not try to all-reduce)
            lr.apply_gradients(0)
    return p0.numpy(), _flat_params(lr).numpy()


def test_data_parallel_update_equals_the_single_process_update_of_the_mean_gradient_gloo_world2():
    """f4: two ranks with DIFFERENT local batches, one all-reduce of the flattened actor + critic gradients per minibatch == a
    single-process update whose gradient is the mean of the two ranks' gradients (computed by hand from two backward passes).  A wrong
    exchange (sum instead of mean, a stale or partial bucket, only one of the two networks) moves the parameters elsewhere: the first
    Adam step is lr * g / (|g| + eps), the clipping thresholds see the gradient norm, and four steps compound it."""
    (_, a0, a1), (_, b0, b1) = _run_world2(same_batch=False)
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)    # broadcast from rank 0; identical updates on both ranks
    r0, r1 = _single_process_reference([_dp_batch(7, 8, _DP_G), _dp_batch(8, 8, _DP_G)])
    assert np.array_equal(r0, a0)
    assert np.abs(a1 - a0).max() > 1e-4                          # four Adam steps of lr 1e-3
    # gloo sums the two float32 buckets in the order the by-hand reference does: bit-equal here; a few ulps would still pass
    assert np.allclose(a1, r1, rtol=0, atol=2e-7), np.abs(a1 - r1).max()
    # and it is NOT what a rank does on its own batch alone, nor with the summed (un-averaged) gradient
    alone0, alone1 = _single_process_reference([_dp_batch(7, 8, _DP_G)])
    assert np.abs(alone1 - a1).max() > 1e-4


def test_data_parallel_update_with_the_same_batch_equals_world1_bit_for_bit():
    """f4: both ranks hold the SAME batch -> the averaged gradient (g + g) / 2 is g exactly, and the world-2 update is the world-1
    update bit for bit."""
    (_, a0, a1), (_, b0, b1) = _run_world2(same_batch=True)
    r0, r1 = _single_process_reference([_dp_batch(7, 8, _DP_G)])
    assert np.array_equal(a0, r0) and np.array_equal(a1, b1)
    assert np.array_equal(a1, r1)


def test_rollout_logp_reproduces_the_stored_log_probabilities_in_the_same_batch_composition():
    """The invariant behind the large first-minibatch approx_kl of a roll-out: with frozen weights, the log-probability of a stored
    action evaluated over the SAME batch the roll-out forward saw is the stored one; over another composition (a minibatch of the
    update) it is not, because the actor runs BatchNorm in training mode (IPPO.py:95-113 has the same property with its batch of one)."""
    import torch
    from multi_agent_rl_wrsn_amd import PPOLearner
    torch.manual_seed(0); torch.set_num_threads(2)
    G, n = 12, 10
    lr = PPOLearner(dict(batch_size=8, minibatch_size=4), 1, G, "cpu", infer_chunk=4)
    formula_fill(lr.actors[0])
    states = torch.rand((n, 4, G, G))
    act, lp = lr.get_action(0, states)                          # chunks of 4, 4, 2 rows
    assert torch.allclose(lr.rollout_logp(0, states, act), lp, rtol=1e-6, atol=1e-4)
    with torch.no_grad():
        other, _ = lr.evaluate(0, states, act)                  # one batch of 10 rows: different BatchNorm statistics
    assert (other - lp).abs().max() > 1e-2


def _rank_probe(path):
    return "import os; open(os.path.join(%r, os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))" % path


def test_launch_ranks_sets_the_torchrun_environment_for_every_child(tmp_path):
    """bench.py / bench_ippo.py `--gpus N` without a launcher: N children with RANK / LOCAL_RANK 0..N-1, WORLD_SIZE N, one common
    MASTER_ADDR / MASTER_PORT; the worst return code comes back."""
    import sys
    from multi_agent_rl_wrsn_amd import launch_ranks
    rc = launch_ranks(3, [sys.executable, "-c", _rank_probe(str(tmp_path))])
    assert rc == 0
    rows = [open(os.path.join(str(tmp_path), str(r))).read().split() for r in range(3)]
    assert [r[0] for r in rows] == ["0", "1", "2"] and [r[1] for r in rows] == ["0", "1", "2"] and all(r[2] == "3" for r in rows)
    assert all(r[3] == "127.0.0.1" for r in rows) and len({r[4] for r in rows}) == 1 and int(rows[0][4]) > 0
    assert launch_ranks(2, [sys.executable, "-c", "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"]) == 3


def test_bench_scripts_start_their_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` / `python bench_ippo.py --gpus 2` with WORLD_SIZE unset hand over to launch_ranks before torch is
    imported (the children here are stubbed by pointing sys.executable's script at a probe through PYTHONSTARTUP-free means: the
    scripts are imported as modules and `launch_ranks` is replaced)."""
    import importlib.util
    import sys
    import multi_agent_rl_wrsn_amd.sharding as sharding
    calls = []
    real = sharding.launch_ranks
    sharding.launch_ranks = lambda n, argv, **kw: calls.append((n, list(argv))) or 0
    env_backup = os.environ.pop("WORLD_SIZE", None)
    try:
        for script in ("bench.py", "bench_ippo.py"):
            spec = importlib.util.spec_from_file_location("probe_" + script[:-3], os.path.join(ROOT, script))
            mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
            argv = sys.argv; sys.argv = [script, "--gpus", "2"]
            had_torch_cuda_init = "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized()
            try:
                with pytest.raises(SystemExit) as ex:
                    mod.main()
            finally:
                sys.argv = argv
            assert ex.value.code == 0
            assert calls[-1][0] == 2 and calls[-1][1][-2:] == ["--gpus", "2"] and calls[-1][1][1].endswith(script)
            if "torch" in sys.modules:
                assert sys.modules["torch"].cuda.is_initialized() == had_torch_cuda_init
    finally:
        sharding.launch_ranks = real
        if env_backup is not None:
            os.environ["WORLD_SIZE"] = env_backup
