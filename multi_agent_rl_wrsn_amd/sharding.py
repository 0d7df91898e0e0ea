"""Multi-GPU sharding of independent environment batches.

Environments never interact, so GPU g of P owns the contiguous range [g*B/P, (g+1)*B/P) with its own handle and
stream and there is no collective on the stepping path.  The one exchange step is the gather of per-environment
rollout statistics (returns per charger, episode counts, lifetimes): a single fused buffer through one
`all_gather` -- RCCL over xGMI with backend "nccl" on MI355X, gloo on CPU for tests.  The reference has no
distributed code at all.
"""
import os


def shard_range(n_total, rank, world):
    """Contiguous, balanced range of environment ids owned by `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).
    Returns (rank, world, local_rank).  A single process (WORLD_SIZE unset or 1) does not create a group."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def launch_ranks(n, argv, stdout_rank0=None):
    """One node, no launcher: start `n` rank processes of `argv` (a command line, e.g. [sys.executable, "bench.py", "--gpus", "8"])
    with the torchrun environment -- RANK / LOCAL_RANK = 0..n-1, WORLD_SIZE = n, MASTER_ADDR 127.0.0.1, a free MASTER_PORT -- wait for
    them and return the worst return code.  Rank 0 inherits stdout (or writes to `stdout_rank0`), the others are silenced.  Must be
    called BEFORE the calling process touches the GPU (it only forks and waits; on this pool a process that initialised HIP must not
    exec another program)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    procs = []
    for r in range(int(n)):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(int(n)), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=(stdout_rank0 if r == 0 else subprocess.DEVNULL)))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class RolloutStats:
    """Per-environment rollout accumulators kept on the shard's device and gathered with ONE collective.

    Layout of the fused buffer, [B_local, M + 3] float64:
        [:, :M]  sum of rewards per charger (the "returns" IPPO consumes: with the reference's `cal_rt_adv`
                 the bootstrap term is multiplied by int(terminal) == 0, so returns == rewards, IPPO.py:80-81)
        [:, M]   finished episodes      [:, M+1]  sum of lifetimes (env.now at terminal)      [:, M+2]  decisions
    """

    def __init__(self, n_env_local, n_agent, device):
        import torch
        self.torch = torch
        self.M = int(n_agent)
        self.buf = torch.zeros((int(n_env_local), self.M + 3), dtype=torch.float64, device=device)

    def update(self, agent_id, reward, terminal, now, status=None):
        """Accumulate one batch of returns.  Pure elementwise device ops (no index kernels, no host sync).
        Rows whose step is still in flight (status 4, VecWRSN step_budget) carry no request and count no decision."""
        t = self.torch
        for m in range(self.M):
            self.buf[:, m] += reward * (agent_id == m)
        term = terminal.to(t.float64)
        self.buf[:, self.M] += term
        self.buf[:, self.M + 1] += term * now
        self.buf[:, self.M + 2] += 1.0 if status is None else (status != 4).to(t.float64)

    def gather(self, group=None):
        return self.gather_table(self.buf, group)

    @staticmethod
    def gather_table(buf, group=None):
        """All ranks receive the [world * B_local, M + 3] table (rank-major, i.e. global environment order for
        equal shards).  One all_gather of 8 * B_local * (M + 3) bytes per rank: latency-bound on xGMI."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return buf.clone()
        world = dist.get_world_size(group)
        out = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
        if hasattr(dist, "all_gather_into_tensor") and buf.is_cuda:
            dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
        else:
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf.contiguous(), group=group)
            out = torch.cat(parts, dim=0)
        return out
