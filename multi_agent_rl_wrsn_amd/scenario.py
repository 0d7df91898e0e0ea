"""Scenario I/O for the vectorised WRSN environment (host side, numpy only).

Mirrors the reference's data formats for the step path:
  * scenario YAML -- `physical_env/network/NetworkIO.py:15-34` (keys node_phy_spe, seed, max_time,
    base_station, nodes, targets; e.g. network_scenarios/hanoi1000n50.yaml:1-21),
  * mobile-charger YAML -- `physical_env/mc/mc_types/default.yaml:2-9` read at `rl_env/WRSN.py:24-25`,
and adds the synthetic network generator BASELINE.json's 200-node / 1000-node configurations need
(no shipped scenario has 200 nodes).
"""
from dataclasses import dataclass, field

import numpy as np

NODE_SPEC_KEYS = ("capacity", "threshold", "com_range", "sen_range", "prob_gp", "package_size", "er", "et", "efs", "emp")
MC_SPEC_KEYS = ("capacity", "threshold", "velocity", "pm", "charging_range", "alpha", "beta", "epsilon")

# Physical parameters of the shipped scenarios (network_scenarios/hanoi1000n200.yaml:1-11) and of the
# shipped charger type (mc_types/default.yaml:2-9).  Data constants, used for synthetic networks.
DEFAULT_NODE_SPEC = dict(capacity=10800.0, threshold=540.0, com_range=80.1, sen_range=40.1, prob_gp=1.0,
                         package_size=400.0, er=1e-4, et=5e-5, efs=1e-8, emp=1.3e-12)
DEFAULT_MC_SPEC = dict(capacity=108000.0, threshold=0.0, velocity=5.0, pm=1.0, charging_range=27.0,
                       alpha=4500.0, beta=30.0, epsilon=1e-10)
DEFAULT_MAX_TIME = 604800.0


@dataclass
class Scenario:
    """One sensor network: what `NetworkIO.makeNetwork` builds its object graph from."""
    node_xy: np.ndarray                 # [N, 2] float64
    target_xy: np.ndarray               # [T, 2] float64
    bs_xy: np.ndarray                   # [2]    float64
    node_spec: dict = field(default_factory=lambda: dict(DEFAULT_NODE_SPEC))
    max_time: float = DEFAULT_MAX_TIME
    seed: int = 0
    name: str = "scenario"

    def __post_init__(self):
        self.node_xy = np.ascontiguousarray(self.node_xy, dtype=np.float64).reshape(-1, 2)
        self.target_xy = np.ascontiguousarray(self.target_xy, dtype=np.float64).reshape(-1, 2)
        self.bs_xy = np.ascontiguousarray(self.bs_xy, dtype=np.float64).reshape(2)
        missing = [k for k in NODE_SPEC_KEYS if k not in self.node_spec]
        if missing:
            raise KeyError("node_phy_spe lacks %s" % missing)
        if float(self.node_spec["prob_gp"]) != 1.0:
            # Node.py:61 draws Python's MT19937 once per alive node per second; every shipped scenario
            # has prob_gp == 1 (the draw is then irrelevant).  prob_gp < 1 is unpinned and unsupported.
            raise ValueError("only prob_gp == 1 is supported (all shipped scenarios)")

    @property
    def n_node(self):
        return self.node_xy.shape[0]

    @property
    def n_target(self):
        return self.target_xy.shape[0]

    def node_spec_array(self):
        return np.array([float(self.node_spec[k]) for k in NODE_SPEC_KEYS], dtype=np.float64)

    def frame(self):
        """[xmin, xmax, ymin, ymax] over nodes and the base station, not targets (Network.py:16-26)."""
        xs = np.concatenate([self.node_xy[:, 0], self.bs_xy[:1]])
        ys = np.concatenate([self.node_xy[:, 1], self.bs_xy[1:]])
        return np.array([xs.min(), xs.max(), ys.min(), ys.max()], dtype=np.float64)


def mc_spec_array(mc_spec):
    return np.array([float(mc_spec[k]) for k in MC_SPEC_KEYS], dtype=np.float64)


def load_scenario_yaml(path):
    """Read a scenario file in the reference's format (NetworkIO.py:15-17, 19-34).

    Raises KeyError('max_time') for the `bacgiang_*` files exactly like the reference does
    (NetworkIO.py:34)."""
    import yaml
    with open(path, "r") as f:
        d = yaml.safe_load(f)
    return Scenario(node_xy=np.array(d["nodes"], dtype=np.float64), target_xy=np.array(d["targets"], dtype=np.float64),
                    bs_xy=np.array(d["base_station"], dtype=np.float64), node_spec=dict(d["node_phy_spe"]),
                    max_time=float(d["max_time"]), seed=int(d["seed"]), name=str(path))


def load_mc_yaml(path):
    """Read a mobile-charger type file (WRSN.py:24-25; mc_types/default.yaml)."""
    import yaml
    with open(path, "r") as f:
        d = yaml.safe_load(f)
    missing = [k for k in MC_SPEC_KEYS if k not in d]
    if missing:
        raise KeyError("mc spec lacks %s" % missing)
    return {k: float(d[k]) for k in MC_SPEC_KEYS}


def scenario_from_golden(z):
    """Scenario + mc spec out of a tests/golden/*.npz fixture (inputs only)."""
    node_spec = {k: float(v) for k, v in zip(NODE_SPEC_KEYS, z["node_spec"])}
    mc_spec = {k: float(v) for k, v in zip(MC_SPEC_KEYS, z["mc_spec"])}
    sc = Scenario(node_xy=z["node_xy"], target_xy=z["target_xy"], bs_xy=z["bs_xy"], node_spec=node_spec,
                  max_time=float(z["max_time"]), seed=int(z["seed"]), name="golden")
    return sc, mc_spec


def synth_scenario(seed, n_node=200, n_target=200, side=None, node_spec=None, max_time=DEFAULT_MAX_TIME):
    """Seeded synthetic network of SURVEY.md section 8d: chain-like relay trees rooted at a base station in
    the middle of a side x side field (shipped scenarios: mean degree 2.05-2.26, routes 22-34 hops), with
    every target inside the sensing range of a node that is connected to the base station at t = 0 --
    otherwise `Network.operate` declares the network dead at t = 0.1 (Network.py:76-77).

    Counter-based RNG (Philox keyed by `seed`), so env e of a batch is reproducible from base_seed + e.
    """
    spec = dict(DEFAULT_NODE_SPEC if node_spec is None else node_spec)
    if side is None:                              # keep the shipped node density: 1000 m for <= 200 nodes
        side = 1000.0 * max(1.0, np.sqrt(n_node / 200.0))
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    com, sen = float(spec["com_range"]), float(spec["sen_range"])
    bs = np.array([side / 2.0, side / 2.0])
    hop_lo, hop_hi = 0.62 * com, 0.995 * com      # U[50, 80) m at the shipped com_range
    min_sep = 0.56 * com                          # keeps chains apart: low degree, long routes
    xy = np.empty((n_node, 2))
    parent = np.full(n_node, -1, dtype=np.int64)
    n = 0
    n_direct = int(rng.integers(2, 5))
    tips = []
    tries = 0
    while n < n_node:
        tries += 1
        if tries > 200000:
            raise RuntimeError("synthetic generator failed to place %d nodes (seed %d)" % (n_node, seed))
        if n < n_direct:
            # direct nodes: inside the base station's reach
            ang = rng.uniform(0, 2 * np.pi)
            r = rng.uniform(0.35 * com, 0.95 * com)
            p = bs + r * np.array([np.cos(ang), np.sin(ang)])
            par = -1
        else:
            # extend a chain tip (chain-like growth) or branch off a random placed node
            if tips and rng.random() < 0.93:
                par = tips[int(rng.integers(0, len(tips)))]
            else:
                par = int(rng.integers(0, n))
            out = xy[par] - bs
            base = np.arctan2(out[1], out[0])
            ang = base + rng.normal(0.0, 0.75)
            r = rng.uniform(hop_lo, hop_hi)
            p = xy[par] + r * np.array([np.cos(ang), np.sin(ang)])
        if p[0] < 0 or p[0] > side or p[1] < 0 or p[1] > side:
            continue
        if n > 0:
            d = np.hypot(xy[:n, 0] - p[0], xy[:n, 1] - p[1])
            if d.min() < min_sep:
                continue
        xy[n] = p
        parent[n] = par
        if par in tips:
            tips.remove(par)
        tips.append(n)
        if len(tips) > 24:
            tips.pop(0)
        n += 1
    # targets: each within 0.93 * sen_range of some node (so it is covered), biased to the outer nodes
    dist_bs = np.hypot(xy[:, 0] - bs[0], xy[:, 1] - bs[1])
    w = 0.25 + dist_bs / dist_bs.max()
    w = w / w.sum()
    owner = rng.choice(n_node, size=n_target, p=w)
    ang = rng.uniform(0, 2 * np.pi, size=n_target)
    r = 0.93 * sen * np.sqrt(rng.uniform(0, 1, size=n_target))
    txy = xy[owner] + np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
    return Scenario(node_xy=xy, target_xy=txy, bs_xy=bs, node_spec=spec, max_time=max_time, seed=int(seed),
                    name="synth_n%d_t%d_s%d" % (n_node, n_target, seed))


def synth_batch(base_seed, n_env, n_node=200, n_target=200, n_unique=None, **kw):
    """List of `n_env` synthetic scenarios, env e seeded base_seed + (e mod n_unique)."""
    n_unique = n_env if n_unique is None else max(1, min(int(n_unique), n_env))
    uniq = [synth_scenario(base_seed + u, n_node, n_target, **kw) for u in range(n_unique)]
    return [uniq[e % n_unique] for e in range(n_env)]
