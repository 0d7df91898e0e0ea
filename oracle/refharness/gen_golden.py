"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the REFERENCE's own classes
(rl_env.WRSN, physical_env.*) in this container on top of the SimPy/gym stand-ins of this directory.

    python oracle/refharness/gen_golden.py [case-name-substring ...]

Each fixture is pure data: the inputs (scenario coordinates and physical parameters read from the
reference's YAML data files or hand-made, the action sequence) and the reference's outputs per
decision (agent id, simulated time, reward, terminal flag, node energy / consumption-rate / status,
charger state, observation).  No reference source text is stored.  The GPU box has no
/root/reference; tests there read only these files.
"""
import os
import sys
import tempfile

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, HERE)
from refload import load_reference, REF_ROOT  # noqa: E402

NODE_KEYS = ("capacity", "threshold", "com_range", "sen_range", "prob_gp", "package_size", "er", "et", "efs", "emp")
MC_KEYS = ("capacity", "threshold", "velocity", "pm", "charging_range", "alpha", "beta", "epsilon")
OBS_FULL_DECISIONS = 3       # full float64 observations kept for the first few decisions
OBS_STRIDE = 7               # strided sample kept for every decision


def scen(name):
    with open(os.path.join(REF_ROOT, "physical_env/network/network_scenarios/%s.yaml" % name)) as f:
        return yaml.safe_load(f)


def mc_default():
    with open(os.path.join(REF_ROOT, "physical_env/mc/mc_types/default.yaml")) as f:
        return yaml.safe_load(f)


def six_node():
    s = dict(scen("hanoi1000n50"))
    s["nodes"] = [[510.0, 505.0], [490.0, 510.0], [560.0, 500.0], [620.0, 520.0], [440.0, 480.0], [380.0, 470.0]]
    s["targets"] = [[640.0, 530.0], [370.0, 460.0], [565.0, 510.0]]
    return s


def frame_of(s):
    xs = [p[0] for p in s["nodes"]] + [s["base_station"][0]]
    ys = [p[1] for p in s["nodes"]] + [s["base_station"][1]]
    return min(xs), max(xs), min(ys), max(ys)


def bs_action(s, tau):
    x0, x1, y0, y1 = frame_of(s)
    return np.array([(s["base_station"][0] - x0) / (x1 - x0), (s["base_station"][1] - y0) / (y1 - y0), tau])


def snapshot(env):
    nodes = env.net.listNodes
    ag = env.agents
    return dict(
        node_energy=np.array([n.energy for n in nodes], dtype=np.float64),
        node_cs=np.array([n.energyCS for n in nodes], dtype=np.float64),
        node_rr=np.array([n.energyRR for n in nodes], dtype=np.float64),
        node_status=np.array([n.status for n in nodes], dtype=np.int32),
        node_level=np.array([(-9 if n.level is None else n.level) for n in nodes], dtype=np.int32),
        mc_energy=np.array([a.energy for a in ag], dtype=np.float64),
        mc_loc=np.array([[float(a.location[0]), float(a.location[1])] for a in ag], dtype=np.float64),
        mc_status=np.array([a.status for a in ag], dtype=np.int32),
        mc_charging=np.array([a.cur_action_type == "charging" for a in ag], dtype=np.int32),
        mc_cur=np.array([[float(v) for v in a.cur_phy_action] for a in ag], dtype=np.float64),
        mc_nconn=np.array([len(a.connected_nodes) for a in ag], dtype=np.int32),
        excl=np.array([float(v) for v in env.agents_exclusive_reward], dtype=np.float64),
        alive=np.int32(env.net.alive),
        targets_active=np.array(env.net.targets_active, dtype=np.int32),
    )


def run_case(name, s, mc, M, actions, max_steps, WRSN, map_size=100, warm_up=100):
    tf = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(s, tf); tf.close()
    tm = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(mc, tm); tm.close()
    env = WRSN(tf.name, tm.name, M, map_size=map_size, warm_up_time=warm_up, density_map=False)
    req = env.reset()
    rec = {k: [] for k in ("in_agent", "in_action", "agent_id", "now", "reward", "terminal", "is_none",
                           "obs_sample")}
    snaps = []
    obs_full = []
    reset_snap = snapshot(env)
    reset_obs = req["state"].copy()
    reset_agent = -1 if req["agent_id"] is None else req["agent_id"]
    for i in range(max_steps):
        aid = req["agent_id"]
        a = np.asarray(actions(i, req), dtype=np.float64)
        req = env.step(aid, a)
        rec["in_agent"].append(-1 if aid is None else aid)
        rec["in_action"].append(a)
        if req is None:      # WRSN.step fell off the end (WRSN.py:321-330 has no else)
            rec["is_none"].append(1); rec["agent_id"].append(-1); rec["now"].append(env.env.now)
            rec["reward"].append(np.nan); rec["terminal"].append(0)
            rec["obs_sample"].append(np.full((4, len(range(0, map_size, OBS_STRIDE)), len(range(0, map_size, OBS_STRIDE))), np.nan))
            snaps.append(snapshot(env))
            break
        rec["is_none"].append(0)
        rec["agent_id"].append(-1 if req["agent_id"] is None else req["agent_id"])
        rec["now"].append(env.env.now)
        rec["reward"].append(np.nan if req["reward"] is None else req["reward"])
        rec["terminal"].append(int(req["terminal"]))
        if req["state"] is not None:
            rec["obs_sample"].append(req["state"][:, ::OBS_STRIDE, ::OBS_STRIDE].copy())
            if len(obs_full) < OBS_FULL_DECISIONS:
                obs_full.append(req["state"].copy())
        else:
            rec["obs_sample"].append(np.full((4, len(range(0, map_size, OBS_STRIDE)), len(range(0, map_size, OBS_STRIDE))), np.nan))
        snaps.append(snapshot(env))
        if req["terminal"]:
            break
    os.unlink(tf.name); os.unlink(tm.name)
    out = dict(
        node_xy=np.array(s["nodes"], dtype=np.float64), target_xy=np.array(s["targets"], dtype=np.float64),
        bs_xy=np.array(s["base_station"], dtype=np.float64),
        node_spec=np.array([float(s["node_phy_spe"][k]) for k in NODE_KEYS]),
        mc_spec=np.array([float(mc[k]) for k in MC_KEYS]),
        max_time=np.float64(s["max_time"]), num_agent=np.int32(M), map_size=np.int32(map_size),
        warm_up=np.float64(warm_up), seed=np.int32(s.get("seed", 0)),
        frame=np.array(env.net.frame, dtype=np.float64),
        consts=np.array([env.moving_time_max, env.charging_time_max, env.avg_nodes_agent, env.net.nodes_density]),
        in_agent=np.array(rec["in_agent"], dtype=np.int32), in_action=np.array(rec["in_action"], dtype=np.float64),
        agent_id=np.array(rec["agent_id"], dtype=np.int32), now=np.array(rec["now"], dtype=np.float64),
        reward=np.array(rec["reward"], dtype=np.float64), terminal=np.array(rec["terminal"], dtype=np.int32),
        is_none=np.array(rec["is_none"], dtype=np.int32),
        obs_sample=np.array(rec["obs_sample"], dtype=np.float64), obs_stride=np.int32(OBS_STRIDE),
        obs_full=np.array(obs_full, dtype=np.float64).reshape(-1, 4, map_size, map_size),
        reset_agent=np.int32(reset_agent), reset_obs=reset_obs,
    )
    for k, v in reset_snap.items():
        out["reset_" + k] = v
    for k in snaps[0]:
        out[k] = np.array([sn[k] for sn in snaps])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-34s decisions=%d end_now=%.3f terminal=%d dead_nodes=%d" % (
        name, len(rec["now"]), rec["now"][-1], rec["terminal"][-1], int((snaps[-1]["node_status"] == 0).sum())), flush=True)


def rnd(seed):
    rng = np.random.RandomState(seed)
    return lambda i, req: rng.rand(3)


def main():
    WRSN, _, _ = load_reference()
    mc = mc_default()
    mc9000 = dict(mc); mc9000["capacity"] = 9000
    mc1500 = dict(mc); mc1500["capacity"] = 1500
    s6 = six_node()
    a_bs = bs_action(s6, 0.01)
    a_bs0 = np.array([a_bs[0], a_bs[1], 0.0])
    corners = [[-1, 2, 0.0], [2, 2, 0.001], [2, 2, 0.0], [0.3, 0.4, 0.002]]
    cases = [
        ("hanoi1000n50_m3_s1", scen("hanoi1000n50"), mc, 3, rnd(1), 40),
        ("hanoi1000n50_m1_s3", scen("hanoi1000n50"), mc, 1, rnd(3), 40),       # BASELINE config #1 shape: 1 env, 1 MC
        ("sonla1000n50_m2_s4", scen("sonla1000n50"), mc, 2, rnd(4), 40),
        ("hanoi1000n100_m3_s5", scen("hanoi1000n100"), mc, 3, rnd(5), 40),
        ("hanoi1000n50_m2_cap9000_detour", scen("hanoi1000n50"), mc9000, 2, rnd(6), 40),
        ("hanoi1000n50_m3_cap1500_mcdeath", scen("hanoi1000n50"), mc1500, 3, rnd(7), 40),
        ("six_m1_bs_charge_ongrid", s6, mc, 1, lambda i, r: a_bs, 12),
        ("six_m3_bs_charge_ongrid", s6, mc, 3, lambda i, r: a_bs, 16),
        ("six_m3_zero_length", s6, mc, 3, lambda i, r: (a_bs0 if i % 3 else a_bs), 16),
        ("six_m2_corners_clipped", s6, mc, 2, lambda i, r: np.array(corners[i % 4], dtype=float), 20),
        ("hanoi1000n150_m3_s9", scen("hanoi1000n150"), mc, 3, rnd(9), 40),
        ("hanoi1000n200_m3_s8", scen("hanoi1000n200"), mc, 3, rnd(8), 40),
    ]
    sel = sys.argv[1:]
    for c in cases:
        if sel and not any(x in c[0] for x in sel):
            continue
        run_case(c[0], c[1], c[2], c[3], c[4], c[5], WRSN)
    # network-only plumbing (runner/test_network.py): state after `run(until=t)` with no charger activity
    if not sel or any("warmup" in x for x in sel):
        for t in (1, 10, 37):
            run_case("hanoi1000n50_m1_warmup%d" % t, scen("hanoi1000n50"), mc, 1, rnd(11), 2, WRSN, warm_up=t)


if __name__ == "__main__":
    main()
