"""The CPU oracle (oracle/wrsn_oracle.c) against every golden fixture generated from the reference
(tests/golden/*.npz, script oracle/refharness/gen_golden.py).  Pins the oracle: float64 on both sides, so the
tolerances here are far tighter than the product's 1e-5."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, oracle_from_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(name):
    z = load_golden(name)
    sc, mc, o = oracle_from_golden(z)
    info = o.env_info()
    assert np.allclose([info["xmin"], info["xmax"], info["ymin"], info["ymax"]], z["frame"], rtol=0, atol=0)
    assert np.allclose([info["moving_time_max"], info["charging_time_max"], info["avg_nodes_agent"], info["nodes_density"]],
                       z["consts"], rtol=1e-14)
    r = o.reset()
    assert (-1 if r["agent_id"] is None else r["agent_id"]) == int(z["reset_agent"])
    nd = o.nodes()
    assert np.allclose(nd["energy"], z["reset_node_energy"], rtol=1e-12)
    assert np.allclose(nd["cs"], z["reset_node_cs"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(nd["status"], z["reset_node_status"])
    assert np.array_equal(nd["level"], z["reset_node_level"])
    assert np.array_equal(o.targets_active(), z["reset_targets_active"])
    assert np.max(np.abs(r["state"] - z["reset_obs"])) < 1e-11
    for k in range(len(z["in_action"])):
        aid = int(z["in_agent"][k])
        r = o.step(None if aid < 0 else aid, z["in_action"][k])
        if z["is_none"][k]:
            assert r["status"] == 1
            break
        assert (-1 if r["agent_id"] is None else r["agent_id"]) == int(z["agent_id"][k]), k
        assert r["terminal"] == bool(z["terminal"][k]), k
        assert abs(r["now"] - z["now"][k]) <= 1e-9 * max(1.0, z["now"][k]), k
        nd = o.nodes(); m = o.mcs()
        assert np.array_equal(nd["status"], z["node_status"][k]), k
        assert np.allclose(nd["energy"], z["node_energy"][k], rtol=1e-10), k
        assert np.allclose(nd["cs"], z["node_cs"][k], rtol=1e-8, atol=1e-11), k
        assert np.allclose(nd["rr"], z["node_rr"][k], rtol=1e-9, atol=1e-12), k
        assert np.array_equal(nd["level"], z["node_level"][k]), k
        assert np.allclose(m["energy"], z["mc_energy"][k], rtol=1e-12, atol=1e-9), k
        assert np.allclose(np.stack([m["loc_x"], m["loc_y"]], 1), z["mc_loc"][k], rtol=1e-12, atol=1e-9), k
        assert np.array_equal(m["status"].astype(int), z["mc_status"][k]), k
        assert np.array_equal(m["type_charging"].astype(int), z["mc_charging"][k]), k
        assert np.array_equal(m["n_conn"].astype(int), z["mc_nconn"][k]), k
        assert np.allclose(m["excl"], z["excl"][k], rtol=1e-9, atol=1e-11), k
        assert np.allclose(np.stack([m["cur_x"], m["cur_y"], m["cur_t"]], 1), z["mc_cur"][k], rtol=1e-12, atol=1e-9), k
        if r["agent_id"] is not None and not r["terminal"]:
            rew = z["reward"][k]
            assert (np.isinf(rew) and r["reward"] == rew) or abs(r["reward"] - rew) <= 1e-9 * max(1e-3, abs(rew)), (k, r["reward"], rew)
            s = int(z["obs_stride"])
            assert np.max(np.abs(r["state"][:, ::s, ::s] - z["obs_sample"][k])) < 1e-10, k
            if k < z["obs_full"].shape[0]:
                assert np.max(np.abs(r["state"] - z["obs_full"][k])) < 1e-10, k
        if r["terminal"]:
            break


def test_oracle_rejects_prob_gp_below_one():
    from wrsn_oracle import OracleWRSN
    from multi_agent_rl_wrsn_amd.scenario import DEFAULT_MC_SPEC, DEFAULT_NODE_SPEC
    spec = dict(DEFAULT_NODE_SPEC); spec["prob_gp"] = 0.5
    with pytest.raises(ValueError):
        OracleWRSN([[510, 505]], [[515, 505]], [500, 500], spec, DEFAULT_MC_SPEC, 1000, 1)
