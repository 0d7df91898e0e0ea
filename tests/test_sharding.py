"""Environment sharding + the one collective of the path (returns all-gather), on CPU with gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch


def test_shard_ranges_partition_the_batch():
    from multi_agent_rl_wrsn_amd import shard_range
    for n, w in ((4096, 8), (10, 3), (7, 8), (32768, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def test_rollout_stats_single_process():
    from multi_agent_rl_wrsn_amd import RolloutStats
    st = RolloutStats(4, 3, "cpu")
    st.update(torch.tensor([0, 2, -1, 1]), torch.tensor([1.0, 2.0, 5.0, -1.0], dtype=torch.float64),
              torch.tensor([0, 0, 1, 0], dtype=torch.uint8), torch.tensor([10.0, 20.0, 30.0, 40.0], dtype=torch.float64))
    g = st.gather()
    assert g.shape == (4, 6)
    assert g[0, 0] == 1.0 and g[1, 2] == 2.0 and g[2, :3].abs().sum() == 0 and g[3, 1] == -1.0
    assert g[2, 3] == 1.0 and g[2, 4] == 30.0 and torch.all(g[:, 5] == 1.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from multi_agent_rl_wrsn_amd import RolloutStats, init_distributed, shard_range
    r, w, _ = init_distributed(backend="gloo")
    lo, hi = shard_range(10, r, w)
    st = RolloutStats(hi - lo, 2, "cpu")
    ids = torch.arange(lo, hi) % 2
    st.update(ids, torch.arange(lo, hi, dtype=torch.float64), torch.zeros(hi - lo, dtype=torch.uint8), torch.zeros(hi - lo, dtype=torch.float64))
    g = st.gather()
    q.put((r, g.numpy()))
    torch.distributed.destroy_process_group()


def test_returns_all_gather_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0], res[1])                    # every rank holds the whole table
    g = res[0]
    assert g.shape == (10, 5)
    for e in range(10):                                      # rank-major order == global environment order
        assert g[e, e % 2] == float(e) and g[e, 1 - e % 2] == 0.0
