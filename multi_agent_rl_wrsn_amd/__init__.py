"""MI355X-native vectorised WRSN environment: the step path of nguyenngocbaocmt02/multi_agent_rl_wrsn
(physical_env.network / physical_env.mc dynamics behind rl_env.WRSN) as hand-written gfx950 HIP kernels behind a
C-ABI (include/wrsn_hip.h).  Python here is host glue: scenario I/O, the batched `VecWRSN`, the drop-in `WRSN`
facade and the environment sharding helpers.  Importing the package does not load the HIP library; constructing an
environment does, and fails loudly when it is missing or no HIP device is present."""
from .scenario import (DEFAULT_MC_SPEC, DEFAULT_NODE_SPEC, Scenario, load_mc_yaml, load_scenario_yaml,  # noqa: F401
                       scenario_from_golden, synth_batch, synth_scenario)
from .sharding import RolloutStats, init_distributed, launch_ranks, shard_range  # noqa: F401
from .vec_env import VecWRSN  # noqa: F401
from .wrsn import WRSN  # noqa: F401
from .ippo import BatchedIPPO, PPOLearner, TransitionBuffers, build_networks, select_batch  # noqa: F401

__all__ = ["Scenario", "load_scenario_yaml", "load_mc_yaml", "synth_scenario", "synth_batch", "scenario_from_golden",
           "DEFAULT_NODE_SPEC", "DEFAULT_MC_SPEC", "VecWRSN", "WRSN", "RolloutStats", "init_distributed", "launch_ranks", "shard_range",
           "BatchedIPPO", "PPOLearner", "TransitionBuffers", "build_networks", "select_batch"]
