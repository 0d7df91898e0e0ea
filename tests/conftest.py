import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emu"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from multi_agent_rl_wrsn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def oracle_from_golden(z):
    from multi_agent_rl_wrsn_amd.scenario import scenario_from_golden
    from wrsn_oracle import OracleWRSN
    sc, mc = scenario_from_golden(z)
    o = OracleWRSN(sc.node_xy, sc.target_xy, sc.bs_xy, sc.node_spec, mc, sc.max_time, int(z["num_agent"]),
                   map_size=int(z["map_size"]), warm_up_time=float(z["warm_up"]))
    return sc, mc, o
