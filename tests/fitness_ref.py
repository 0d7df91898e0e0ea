"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the reference's `WRSN.get_network_fitness`
(rl_env/WRSN.py:188-220) and `get_reward` (:222-227), evaluated on an arbitrary node state.

Used where a reward of the implementation under test may legitimately differ from the fixture: the reference divides
by `energyCS`, and a node that has been idle for 10 s keeps the rounding residue of its sliding mean (+-1e-16 instead of
0, Node.py:71-77), whose SIGN decides whether the node is a bottleneck.  In that case the tests check
that (a) the node state matches the fixture and (b) the implementation's fitness / reward are exactly what the
reference's algorithm yields on the implementation's OWN node state."""
import numpy as np


class Topology:
    """Neighbour / coverage / direct-node lists as the reference probes them (Node.py:80-90, BaseStation.py:20-23)."""

    def __init__(self, node_xy, target_xy, bs_xy, com_range, sen_range):
        xy = np.asarray(node_xy, dtype=np.float64); N = len(xy)
        d = np.sqrt((xy[:, None, 0] - xy[None, :, 0]) ** 2 + (xy[:, None, 1] - xy[None, :, 1]) ** 2)
        self.neighbors = [[j for j in range(N) if j != i and d[i, j] <= com_range] for i in range(N)]
        self.direct = [i for i in range(N) if np.sqrt((xy[i, 0] - bs_xy[0]) ** 2 + (xy[i, 1] - bs_xy[1]) ** 2) <= com_range]
        t = np.asarray(target_xy, dtype=np.float64)
        dt = np.sqrt((xy[:, None, 0] - t[None, :, 0]) ** 2 + (xy[:, None, 1] - t[None, :, 1]) ** 2)
        self.covered = [[q for q in range(len(t)) if dt[i, q] <= sen_range] for i in range(N)]
        self.n_target = len(t)


def network_fitness(topo, energy, cs, status, threshold):
    """WRSN.py:188-220, statement for statement; returns target_t."""
    N = len(energy)
    node_t = [-1.0] * N
    tmp1 = []
    for i in topo.direct:
        if status[i] == 1:
            tmp1.append(i)
            node_t[i] = float("inf") if cs[i] == 0 else (energy[i] - threshold) / cs[i]
    while tmp1:
        tmp2 = []
        for i in tmp1:
            for j in topo.neighbors[i]:
                if status[j] != 1:
                    continue
                lt = float("inf") if cs[j] == 0 else (energy[j] - threshold) / cs[j]
                if node_t[j] == -1 or (node_t[i] > node_t[j] and lt > node_t[j]):
                    tmp2.append(j)
                    node_t[j] = min(lt, node_t[i])
        tmp1 = tmp2
    target_t = [0.0] * topo.n_target
    for i in range(N):
        for q in topo.covered[i]:
            target_t[q] = max(target_t[q], node_t[i])
    return np.array(target_t)


def reward(min_fit, prev_min_fit, excl, avg_nodes_agent, charging_time_max, moving_time_max):    # WRSN.py:222-227
    return ((min_fit - prev_min_fit) * 0.8 + 0.2 * (excl / avg_nodes_agent)) / (charging_time_max + moving_time_max)


def residue_nodes(cs, status):
    """alive nodes whose consumption rate is a rounding residue (non-zero, 1e-9 below the largest rate)"""
    cs = np.asarray(cs, dtype=np.float64); alive = np.asarray(status) == 1
    scale = max(float(np.abs(cs).max()), 1e-30)
    return alive & (cs != 0) & (np.abs(cs) < 1e-9 * scale)
