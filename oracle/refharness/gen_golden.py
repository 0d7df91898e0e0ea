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


def redundant_net(capacity=1000.0, max_time=None):
    """Hand-made network with redundant coverage, so that nodes die WITHOUT ending the episode (Node.py:92-132,
    Network.py:69-81).  A hub of 12 mutually adjacent nodes next to the base station (every hub node has > 8
    neighbours, the hub targets have > 8 covering nodes) relays the packets of source sites further west; a hub node
    that carries a site spends 2.5 x what the sources spend, runs dry first and the site falls back on the next hub
    node at once (Node.py:93-94 tests the live status).  Sites come in mirror-image pairs (their carriers run dry in the
    same second) and every site is two nodes that cover the same targets.  To the east a ladder of two cross-linked
    chains (a rung target is covered by both chains; when a chain's head dies the other takes over through the cross
    links after the next setLevels).  Low node capacity: all of this happens within a few decisions."""
    s = dict(scen("hanoi1000n50"))
    spec = dict(s["node_phy_spe"]); spec["capacity"] = capacity
    s["node_phy_spe"] = spec
    if max_time is not None:
        s["max_time"] = max_time
    rng = np.random.RandomState(12345)
    nodes, targets = [], []
    hub_c = (440.0, 500.0)
    for k in range(6):                                        # hub: 6 mirror pairs inside a 24 m disc
        r = 6.0 + 18.0 * rng.rand(); a = np.pi * (0.08 + 0.84 * rng.rand())
        dx, dy = float(np.round(r * np.cos(a), 1)), float(np.round(abs(r * np.sin(a)) + 1.0, 1))
        nodes.append([hub_c[0] + dx, hub_c[1] + dy]); nodes.append([hub_c[0] + dx, hub_c[1] - dy])
    for t in range(3):
        targets.append([hub_c[0] + 3.0 * t - 3.0, hub_c[1]])
    for (ox, oy, nt) in ((-58.0, 30.0, 14), (-30.0, 62.0, 10)):   # source sites (mirrored in y), two nodes each
        for sign in (+1.0, -1.0):
            cx, cy = hub_c[0] + ox, hub_c[1] + sign * oy
            nodes.append([cx, cy]); nodes.append([cx - 9.0, cy + sign * 12.0])
            for t in range(nt):
                targets.append([cx - 5.0 - 1.5 * t, cy + sign * (4.0 + 0.5 * t)])
    for k in range(5):                                        # ladder to the east: chains A (y = 521) and B (y = 481)
        x = 561.0 + 70.0 * k
        nodes.append([x, 521.0]); nodes.append([x, 481.0])
        targets.append([x - 5.0, 501.0])
    s["nodes"] = nodes; s["targets"] = targets
    return s


def synth_tree(seed, n_node, n_target):
    """Seeded chain-like relay trees around the base station (the shape of the shipped scenarios: mean degree ~2.2),
    every target inside the sensing range of a node: a 300-node network for the multi-register-slot kernels."""
    s = dict(scen("hanoi1000n50"))
    rng = np.random.RandomState(seed)
    com, sen = float(s["node_phy_spe"]["com_range"]), float(s["node_phy_spe"]["sen_range"])
    side = 1000.0 * max(1.0, np.sqrt(n_node / 200.0)); bs = np.array([side / 2, side / 2])
    pts = []; tips = []
    while len(pts) < n_node:
        if len(pts) < 3:
            a = rng.uniform(0, 2 * np.pi); r = rng.uniform(0.35 * com, 0.95 * com); p = bs + r * np.array([np.cos(a), np.sin(a)]); par = -1
        else:
            par = tips[rng.randint(len(tips))] if (tips and rng.rand() < 0.93) else rng.randint(len(pts))
            o = pts[par] - bs; a = np.arctan2(o[1], o[0]) + 0.75 * rng.randn(); r = rng.uniform(0.62 * com, 0.995 * com)
            p = pts[par] + r * np.array([np.cos(a), np.sin(a)])
        if p.min() < 0 or p.max() > side or any(np.hypot(*(q - p)) < 0.56 * com for q in pts):
            continue
        if par in tips: tips.remove(par)
        pts.append(p); tips.append(len(pts) - 1); tips[:] = tips[-24:]
    tg = []
    for _ in range(n_target):
        o = pts[rng.randint(n_node)]; a = rng.uniform(0, 2 * np.pi); r = 0.93 * sen * np.sqrt(rng.rand())
        tg.append(o + r * np.array([np.cos(a), np.sin(a)]))
    s["base_station"] = [float(bs[0]), float(bs[1])]
    s["nodes"] = [[float(np.round(p[0], 3)), float(np.round(p[1], 3))] for p in pts]
    s["targets"] = [[float(np.round(p[0], 3)), float(np.round(p[1], 3))] for p in tg]
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


class MinimizeRecorder:
    """Wraps the `minimize` name the reference's WRSN module imported from SciPy (WRSN.py:8, :249) and keeps what the
    reference itself passed and got back: start point, bounds, the optimiser's result x, and the value of the
    reference's own objective_function at the start point and at x (the reference only uses x; `res.fun` is kept too --
    after a failed line search on this discontinuous objective it is not always the value at x)."""

    def __init__(self, module):
        self.module = module; self.orig = module.minimize; self.calls = []
        module.minimize = self

    def __call__(self, fun, x0, **kw):
        res = self.orig(fun, x0, **kw)
        self.calls.append(dict(x0=np.array(x0, dtype=np.float64), bounds=np.array(kw.get("bounds"), dtype=np.float64),
                               x=np.array(res.x, dtype=np.float64), fun=float(fun(res.x)), fun_x0=float(fun(x0)), fun_reported=float(res.fun)))
        return res

    def close(self):
        self.module.minimize = self.orig


def run_case(name, s, mc, M, actions, max_steps, WRSN, map_size=100, warm_up=100, density_map=False):
    tf = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(s, tf); tf.close()
    tm = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(mc, tm); tm.close()
    env = WRSN(tf.name, tm.name, M, map_size=map_size, warm_up_time=warm_up, density_map=density_map)
    recorder = None
    if density_map:
        import rl_env.WRSN as wrsn_module
        recorder = MinimizeRecorder(wrsn_module)
    dm = {k: [] for k in ("in_map", "dm_x0", "dm_bounds", "dm_x", "dm_fun", "dm_fun_x0", "dm_fun_reported")}
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
        if density_map:
            # the policy's G x G map, made exactly representable in float32 (the fixture stores it as such); in_action is
            # the 3-vector the reference itself derived from it (WRSN.py:293-299), so that the physics of the fixture
            # can be replayed through the 3-vector path
            a = a.astype(np.float32).astype(np.float64)
            n0 = len(recorder.calls)
        req = env.step(aid, a)
        rec["in_agent"].append(-1 if aid is None else aid)
        if density_map:
            c = recorder.calls[n0]; assert len(recorder.calls) == n0 + 1
            dm["in_map"].append(a.astype(np.float32))
            for k in ("x0", "bounds", "x", "fun", "fun_x0", "fun_reported"):
                dm["dm_" + k].append(c[k])
            a = np.array(env.agents_action[aid], dtype=np.float64)
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
    if recorder is not None:
        recorder.close()
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
    if density_map:
        out["in_map"] = np.array(dm["in_map"], dtype=np.float32)
        for k in ("dm_x0", "dm_bounds", "dm_x", "dm_fun", "dm_fun_x0", "dm_fun_reported"):
            out[k] = np.array(dm[k], dtype=np.float64)
    for k, v in reset_snap.items():
        out["reset_" + k] = v
    for k in snaps[0]:
        out[k] = np.array([sn[k] for sn in snaps])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-34s decisions=%d end_now=%.3f terminal=%d dead_nodes=%s" % (
        name, len(rec["now"]), rec["now"][-1], rec["terminal"][-1], [int((sn["node_status"] == 0).sum()) for sn in snaps]), flush=True)


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
    # ---- round 2: semantics the cases above never reach
    red = redundant_net()
    def short(seed, tmax):
        rng = np.random.RandomState(seed)
        return lambda i, req: np.array([rng.rand(), rng.rand(), tmax * rng.rand()])
    extra = [
        # nodes die without ending the episode, several in one second, re-levelling, > 8 neighbours / covering nodes
        ("redundant_m2_deaths", red, mc, 2, short(21, 0.6), 60, {}),
        ("redundant_m3_deaths", red, mc, 3, short(22, 1.0), 60, {}),
        # Network.operate stops at max_time (Network.py:78-79): levels and `alive` freeze, steps continue, nodes die
        ("redundant_m2_maxtime130", redundant_net(max_time=130), mc, 2, short(23, 0.8), 24, {}),
        ("hanoi1000n50_m3_maxtime400", dict(scen("hanoi1000n50"), max_time=400), mc, 3, rnd(24), 12, {}),
        # 64 x 64 observation
        ("hanoi1000n50_m2_map64", scen("hanoi1000n50"), mc, 2, rnd(25), 12, dict(map_size=64)),
        ("redundant_m2_map64", red, mc, 2, short(26, 0.5), 16, dict(map_size=64)),
        # 300 nodes: two register slots per lane are not enough (NPL = 8 kernels), CSR neighbour lists
        ("synth300_m3_s27", synth_tree(27, 300, 150), mc, 3, short(27, 0.05), 7, {}),
    ]
    for name, sc_, mc_, M_, pol, n_, kw in extra:
        if sel and not any(x in name for x in sel):
            continue
        run_case(name, sc_, mc_, M_, pol, n_, WRSN, **kw)
    # ---- density_map=True (WRSN.py:229-297, what runner/IPPO.py and runner/checkRL.py use): the policy emits a G x G map
    def map_policy(seed, G):
        rng = np.random.RandomState(seed)
        def pol(i, req):
            kind = i % 5
            if kind == 0:                                      # logits
                return rng.randn(G, G) * (1.0 + (i // 5) % 3)
            if kind == 1:                                      # the heuristic of controller/random/RandomController.py:15
                st = req["state"]
                return st[0] + st[1] - 10 * st[2] + st[3]
            if kind == 2:                                      # a peaked map of non-negative weights (goes through exp)
                m = rng.rand(G, G) ** 8
                return m / m.sum()
            if kind == 3:                                      # flat: ties at the percentile and at the arg-max
                return np.full((G, G), 0.5)
            m = np.zeros((G, G)); m[rng.randint(G), rng.randint(G)] = 1.0     # one-hot: a valid probability map
            return m
        return pol
    dens = [
        ("hanoi1000n50_m3_density", scen("hanoi1000n50"), mc, 3, map_policy(31, 100), 20, dict(density_map=True)),
        ("redundant_m2_density_map64", red, mc, 2, map_policy(32, 64), 20, dict(density_map=True, map_size=64)),
    ]
    for name, sc_, mc_, M_, pol, n_, kw in dens:
        if sel and not any(x in name for x in sel):
            continue
        run_case(name, sc_, mc_, M_, pol, n_, WRSN, **kw)
    # network-only plumbing (runner/test_network.py): state after `run(until=t)` with no charger activity
    if not sel or any("warmup" in x for x in sel):
        for t in (1, 10, 37):
            run_case("hanoi1000n50_m1_warmup%d" % t, scen("hanoi1000n50"), mc, 1, rnd(11), 2, WRSN, warm_up=t)


if __name__ == "__main__":
    main()
