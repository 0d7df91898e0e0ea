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


def check_decision(z, k, got, where=""):
    """`got`: dict with agent_id, now, reward, terminal and (optional) node/mc arrays + obs for decision k."""
    tag = "%s decision %d" % (where, k)
    exp_id = int(z["agent_id"][k])
    assert got["agent_id"] == exp_id, (tag, "agent", got["agent_id"], exp_id)
    assert bool(got["terminal"]) == bool(z["terminal"][k]), (tag, "terminal")
    assert close(got["now"], z["now"][k], rtol=1e-9), (tag, "now", got["now"], float(z["now"][k]))
    terminal = bool(z["terminal"][k])
    if exp_id >= 0:
        r = float(z["reward"][k])
        assert close(got["reward"], r, atol=1e-9), (tag, "reward", got["reward"], r)
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
    if got.get("obs") is not None and exp_id >= 0 and not terminal:
        s = int(z["obs_stride"])
        ref = z["obs_sample"][k]
        sample = np.asarray(got["obs"], dtype=np.float64)[:, ::s, ::s]
        scale = max(1.0, float(np.nanmax(np.abs(ref))))
        assert np.max(np.abs(sample - ref)) <= 1e-5 * scale, (tag, "obs sample", np.max(np.abs(sample - ref)), scale)
        if k < z["obs_full"].shape[0]:
            full = z["obs_full"][k]
            assert np.max(np.abs(np.asarray(got["obs"], dtype=np.float64) - full)) <= 1e-5 * max(1.0, float(np.abs(full).max())), (tag, "obs full")
