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

    The generator itself is host code behind the C-ABI (`wrsn_synth_network`, csrc/wrsn_api.hip; its own
    xoshiro256** stream keyed by `seed`), so env e of a batch is reproducible from base_seed + e and 4096
    networks take well under a second.  side=None keeps the shipped node density (1000 m for <= 200 nodes).
    """
    from . import _lib
    spec = dict(DEFAULT_NODE_SPEC if node_spec is None else node_spec)
    lib = _lib.load()
    node_xy = np.zeros((int(n_node), 2)); target_xy = np.zeros((int(n_target), 2)); bs = np.zeros(2)
    _lib.check(lib, lib.wrsn_synth_network(int(seed), int(n_node), int(n_target), float(side or 0.0),
                                           float(spec["com_range"]), float(spec["sen_range"]),
                                           node_xy.ctypes.data, target_xy.ctypes.data, bs.ctypes.data))
    return Scenario(node_xy=node_xy, target_xy=target_xy, bs_xy=bs, node_spec=spec, max_time=max_time, seed=int(seed),
                    name="synth_n%d_t%d_s%d" % (n_node, n_target, seed))


def synth_batch(base_seed, n_env, n_node=200, n_target=200, n_unique=None, **kw):
    """List of `n_env` synthetic scenarios, env e seeded base_seed + (e mod n_unique)."""
    n_unique = n_env if n_unique is None else max(1, min(int(n_unique), n_env))
    uniq = [synth_scenario(base_seed + u, n_node, n_target, **kw) for u in range(n_unique)]
    return [uniq[e % n_unique] for e in range(n_env)]
