"""C-ABI surface of libwrsn_hip.so: loads on a CPU-only box, exports every symbol include/wrsn_hip.h declares,
refuses to run without a HIP device (no CPU fallback), and its host-only generator works."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "wrsn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wrsn_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree(hip_lib):
    from multi_agent_rl_wrsn_amd import _lib
    declared = _declared_functions()
    assert declared == sorted(_lib.EXPORTS), (declared, sorted(_lib.EXPORTS))
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert b"gfx950" in hip_lib.wrsn_version()


def test_library_contains_gfx950_code_object():
    from multi_agent_rl_wrsn_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"wrsn_step_kernel" in blob and b"wrsn_warmup_kernel" in blob


def test_struct_layouts_match_header(hip_lib):
    from multi_agent_rl_wrsn_amd import _lib
    assert C.sizeof(_lib.WrsnCfg) == 8 * 4 + 8
    assert C.sizeof(_lib.WrsnNodeSpec) == 11 * 8 and C.sizeof(_lib.WrsnMcSpec) == 8 * 8
    assert C.sizeof(_lib.WrsnStepOut) == 6 * C.sizeof(C.c_void_p)


def test_create_without_gpu_fails_loudly(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from multi_agent_rl_wrsn_amd import _lib
    with pytest.raises(_lib.WrsnError) as ei:
        _lib.RawHandle(hip_lib, 1, 10, 10, 1)
    assert ei.value.code in (-3, -2)          # WRSN_ERR_NO_DEVICE (or a HIP runtime error on exotic hosts)
    from multi_agent_rl_wrsn_amd import DEFAULT_MC_SPEC, VecWRSN, synth_scenario
    with pytest.raises(RuntimeError):
        VecWRSN([synth_scenario(1, 20, 10)], DEFAULT_MC_SPEC, 1)


def test_bad_arguments_are_rejected(hip_lib):
    from multi_agent_rl_wrsn_amd import _lib
    cfg = _lib.WrsnCfg(0, 10, 10, 1, 100, 0, 0, 0, 100.0)
    h = C.c_void_p()
    assert hip_lib.wrsn_create(C.byref(cfg), C.byref(h)) == -1 and b"wrsn_cfg" in hip_lib.wrsn_last_error()
    cfg = _lib.WrsnCfg(1, 10, 10, 9, 100, 0, 0, 0, 100.0)      # more chargers than WRSN_MAX_MC
    assert hip_lib.wrsn_create(C.byref(cfg), C.byref(h)) == -1
    assert hip_lib.wrsn_reset(None, None, None) == -1


def test_synthetic_generator_is_deterministic_and_connected(hip_lib):
    from multi_agent_rl_wrsn_amd import DEFAULT_MC_SPEC, synth_scenario
    from wrsn_oracle import OracleWRSN
    a = synth_scenario(42, 200, 200); b = synth_scenario(42, 200, 200); c = synth_scenario(43, 200, 200)
    assert np.array_equal(a.node_xy, b.node_xy) and np.array_equal(a.target_xy, b.target_xy)
    assert not np.array_equal(a.node_xy, c.node_xy)
    assert a.node_xy.min() >= 0 and a.node_xy.max() <= 1000
    for s in (a, c, synth_scenario(5, 1000, 1000)):
        o = OracleWRSN(s.node_xy, s.target_xy, s.bs_xy, s.node_spec, DEFAULT_MC_SPEC, s.max_time, 1, warm_up_time=2)
        r = o.reset(with_state=False)
        assert not r["terminal"], "every target must be covered by a node connected to the base station"
        t = o.topology()
        assert t["degree"].mean() < 6 and t["n_cover"].sum() >= s.n_target
