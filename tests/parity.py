"""Shared comparison helpers: golden fixture / oracle versus an implementation under test.

Tolerances (BASELINE.json north_star: node-energy and reward trajectories within 1e-5 relative):
  * agent id, terminal flag, node status, charger status / action type: exact;
  * simulated time: 1e-9 relative (float64 on both sides);
  * node energy, consumption rate, charger energy / position, reward: RTOL = 1e-5 (+ tiny absolute floors);
  * observation (float32 on the device): 1e-5 of the map's peak value, absolute.
"""
import numpy as np

RTOL = 1e-5


def close(a, b, rtol=RTOL, atol=0.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    return bool(np.all(ok | both_nan | both_inf))


_TOPO = {}


def _topology(z):
    import fitness_ref
    key = (z["node_xy"].tobytes(), z["target_xy"].tobytes())
    if key not in _TOPO:
        _TOPO[key] = fitness_ref.Topology(z["node_xy"], z["target_xy"], z["bs_xy"], float(z["node_spec"][2]), float(z["node_spec"][3]))
    return _TOPO[key]


def _reward_depends_on_residue(z, k, got, tag):
    """The fixture's reward and the implementation's differ.  Accept it only when the reference's own arithmetic is
    ill-conditioned there -- an alive node carries a rounding-residue energyCS (fitness_ref docstring) -- AND the node
    state matches the fixture AND the implementation's fitness and reward are exactly what the reference's algorithm
    (WRSN.py:188-227) gives on the implementation's own node state."""
    import fitness_ref
    if "node_cs" not in got or "min_fitness" not in got:
        return False
    noisy = fitness_ref.residue_nodes(z["node_cs"][k], z["node_status"][k]) | fitness_ref.residue_nodes(got["node_cs"], got["node_status"])
    if not noisy.any():
        return False
    thr = float(z["node_spec"][1])
    fit = fitness_ref.network_fitness(_topology(z), got["node_energy"], got["node_cs"], got["node_status"], thr)
    assert close(got["min_fitness"], fit.min(), rtol=1e-9), (tag, "fitness on own state", got["min_fitness"], fit.min())
    a = got["agent_id"]
    mtm, ctm, avg = float(z["consts"][0]), float(z["consts"][1]), float(z["consts"][2])
    want = fitness_ref.reward(fit.min(), got["prev_minfit"][a], got["excl"][a], avg, ctm, mtm)
    scale = (0.8 * abs(fit.min() - got["prev_minfit"][a]) + 0.2 * abs(got["excl"][a]) / avg) / (ctm + mtm)
    assert abs(got["reward"] - want) <= 1e-5 * max(abs(want), scale) + 1e-12, (tag, "reward on own state", got["reward"], want)
    return True


def check_decision(z, k, got, where="", noise=None):
    """`got`: dict with agent_id, now, reward, terminal and (optional) node/mc arrays + obs for decision k.
    `noise`: list that collects the decisions whose reward hangs on the sign of a rounding-residue energyCS (see
    _reward_depends_on_residue); without it such a decision fails like any other mismatch."""
    tag = "%s decision %d" % (where, k)
    exp_id = int(z["agent_id"][k])
    assert got["agent_id"] == exp_id, (tag, "agent", got["agent_id"], exp_id)
    assert bool(got["terminal"]) == bool(z["terminal"][k]), (tag, "terminal")
    assert close(got["now"], z["now"][k], rtol=1e-9), (tag, "now", got["now"], float(z["now"][k]))
    terminal = bool(z["terminal"][k])
    if "targets_active" in got:                             # Network.targets_active: what the last setLevels reached (also at the terminal return)
        nt = len(z["targets_active"][k])
        assert np.array_equal(np.asarray(got["targets_active"]).astype(int)[:nt], z["targets_active"][k]), (tag, "targets_active")
    reward_ok = True
    if exp_id >= 0:
        r = float(z["reward"][k])
        reward_ok = close(got["reward"], r, atol=1e-9)
    if "node_energy" in got and not terminal:
        # after the network is declared dead the product freezes node state (documented deviation)
        assert np.array_equal(got["node_status"], z["node_status"][k]), (tag, "node status")
        assert close(got["node_energy"], z["node_energy"][k]), (tag, "node energy", np.max(np.abs(got["node_energy"] - z["node_energy"][k]) / z["node_energy"][k]))
        assert close(got["node_cs"], z["node_cs"][k], atol=1e-9), (tag, "node cs")
    if "mc_energy" in got and not terminal:
        assert close(got["mc_energy"], z["mc_energy"][k], atol=1e-6), (tag, "mc energy")
        assert close(got["mc_loc"], z["mc_loc"][k], atol=1e-6), (tag, "mc loc")
        assert np.array_equal(np.asarray(got["mc_status"]).astype(int), z["mc_status"][k]), (tag, "mc status")
        assert np.array_equal(np.asarray(got["mc_charging"]).astype(int), z["mc_charging"][k]), (tag, "mc action type")
        assert np.array_equal(np.asarray(got["mc_nconn"]).astype(int), z["mc_nconn"][k]), (tag, "connected nodes")
        assert close(got["excl"], z["excl"][k], atol=1e-7), (tag, "exclusive reward", got["excl"], z["excl"][k])
    if not reward_ok:
        if noise is not None and not terminal and _reward_depends_on_residue(z, k, got, tag):
            noise.append((where, k))
        else:
            raise AssertionError((tag, "reward", got["reward"], float(z["reward"][k])))
    if got.get("obs") is not None and exp_id >= 0 and not terminal:
        s = int(z["obs_stride"])
        ref = z["obs_sample"][k]
        sample = np.asarray(got["obs"], dtype=np.float64)[:, ::s, ::s]
        scale = max(1.0, float(np.nanmax(np.abs(ref))))
        assert np.max(np.abs(sample - ref)) <= 1e-5 * scale, (tag, "obs sample", np.max(np.abs(sample - ref)), scale)
        if k < z["obs_full"].shape[0]:
            full = z["obs_full"][k]
            assert np.max(np.abs(np.asarray(got["obs"], dtype=np.float64) - full)) <= 1e-5 * max(1.0, float(np.abs(full).max())), (tag, "obs full")


def check_density_action(z, k, act, nodes, where=""):
    """Fixture with `density_map=True` (the reference ran WRSN.step on G x G policy maps, WRSN.py:293-297, 229-287):
    `act` is what the implementation derived from map k on the node state `nodes` (energy / cs / status of the decision
    before: the state the reference optimised on).  Pinned by the fixture: third component (exact arithmetic on the map:
    arg-max value / mass above the 99.9th percentile), the box of the arg-max cell, and the objective value, which must
    not be below what SciPy's L-BFGS-B reached inside the reference (the spot itself is not reproducible across SciPy
    versions).  tests/density_ref.objective is itself pinned here against the reference's objective_function at the two
    points the fixture holds (box centre and the optimiser's result)."""
    import density_ref
    tag = "%s decision %d (density map)" % (where, k)
    frame = z["frame"]; W, H = frame[1] - frame[0], frame[3] - frame[2]
    ref3 = z["in_action"][k]                                   # the reference's 3-vector after np.clip (WRSN.py:299)
    assert abs(act[2] - ref3[2]) <= 1e-12 * max(ref3[2], 1e-300) or (ref3[2] == 1.0 and act[2] >= 1.0), (tag, "third", act[2], ref3[2])
    (lx, ux), (ly, uy) = z["dm_bounds"][k]
    spot = np.array([act[0] * W + frame[0], act[1] * H + frame[2]])
    tol = 1e-9 * max(ux - lx, uy - ly)
    assert lx - tol <= spot[0] <= ux + tol and ly - tol <= spot[1] <= uy + tol, (tag, "box", spot, z["dm_bounds"][k])
    args = (z["node_xy"], nodes["status"] == 1, nodes["energy"], nodes["cs"], float(z["node_spec"][1]), float(z["mc_spec"][4]),
            float(z["mc_spec"][5]), float(z["mc_spec"][6]))
    ref_best = -float(z["dm_fun"][k])
    # the checker's objective == the reference's own objective_function (values recorded inside the reference run)
    assert close(density_ref.objective(z["dm_x0"][k], *args), -float(z["dm_fun_x0"][k]), rtol=1e-7, atol=1e-12), (tag, "objective at the box centre")
    assert close(density_ref.objective(z["dm_x"][k], *args), ref_best, rtol=1e-7, atol=1e-12), (tag, "objective at SciPy's optimum")
    mine = density_ref.objective(np.clip(spot, [lx, ly], [ux, uy]), *args)
    assert mine >= ref_best * (1 - 1e-7) - 1e-12, (tag, "objective", mine, ref_best)
    return mine, ref_best
