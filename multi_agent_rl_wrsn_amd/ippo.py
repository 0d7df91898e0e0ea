"""Batched independent-PPO roll-out and training on top of `VecWRSN` (f2 roll-out glue, f4 data-parallel
training).  Counterpart of the reference's `controller/ippo/IPPO.py` for B environments per GPU instead of one:

  reference (one environment, Python lists)                       here (B environments, device buffers)
  ------------------------------------------------------------    ------------------------------------------------------
  roll_out bookkeeping            IPPO.py:119-156                  `TransitionBuffers` + HIP kernels behind the C-ABI
                                                                   (`wrsn_rollout_record` / `wrsn_rollout_collect`)
  first transition dropped        IPPO.py:146-147                  a charger without a pending action appends nothing
  cal_rt_adv                      IPPO.py:71-93                    `BatchedIPPO.cal_rt_adv` (stored terminals are all False,
                                                                   so returns == rewards, advantages = r - V(s))
  outlier batch selection         IPPO.py:193-209                  `select_batch` (same index arithmetic, same numpy calls)
  UNet actor / CNN critic         controller/ppo/actor/UnetActor.py:61-80, critic/CNNCritic.py:7-49   `UNet`, `CNNCritic`
                                                                   (same parameter names and shapes: reference
                                                                   checkpoints load with load_state_dict)
  PPO update                      IPPO.py:225-271                  `BatchedIPPO.update` (+ one all-reduce of the flattened
                                                                   actor/critic gradients per minibatch when data-parallel)

torch is used for the policy networks and for device memory; everything that touches environment state goes through
the C-ABI.  What differs from the reference by construction: actions are sampled for all environments of a launch that
carry a request for that charger in one forward pass, so BatchNorm statistics are taken over that batch (the reference
runs the actor on batches of one, in training mode); logging goes to CSV (tensorboard is not a dependency).
"""
import math
import os
import time

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


# ------------------------------------------------------------------------------------------------------------------
# policy networks (PyTorch): same architecture, parameter names and initialisation as the reference's
# ------------------------------------------------------------------------------------------------------------------
def _ortho(layer, std=math.sqrt(2.0), bias=0.0):
    """utils.py:19-22 layer_init: orthogonal weights, constant bias."""
    torch = _torch()
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias)
    return layer


def build_networks(map_size=100):
    """Returns (UNet, CNNCritic) classes bound to `map_size` (the reference hard-codes 100 x 100)."""
    torch = _torch()
    nn = torch.nn
    F = torch.nn.functional

    class _Block(nn.Module):                                 # conv 3x3 + BatchNorm + ReLU   (UnetActor.py:6-17)
        def __init__(self, cin, cout):
            super().__init__()
            self.conv = _ortho(nn.Conv2d(cin, cout, kernel_size=3, padding=1))
            self.bn = nn.BatchNorm2d(cout)

        def forward(self, x):
            return F.relu(self.bn(self.conv(x)), inplace=True)

    class _Down(nn.Module):                                  # max-pool 2 then block          (UnetActor.py:19-30)
        def __init__(self, cin, cout):
            super().__init__()
            self.conv_block = _Block(cin, cout)

        def forward(self, x):
            return self.conv_block(F.max_pool2d(x, 2))

    class _Up(nn.Module):                                    # bilinear x2, pad to the skip, concat, block   (UnetActor.py:33-47)
        def __init__(self, cin, cout):
            super().__init__()
            self.conv_block = _Block(cin, cout)

        def forward(self, low, skip):
            low = F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)
            dy, dx = skip.shape[2] - low.shape[2], skip.shape[3] - low.shape[3]
            low = F.pad(low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
            return self.conv_block(torch.cat([skip, low], dim=1))

    class _Out(nn.Module):                                   # conv 3x3, small init            (UnetActor.py:50-58)
        def __init__(self, cin, cout):
            super().__init__()
            self.conv = _ortho(nn.Conv2d(cin, cout, kernel_size=3, padding=1), std=0.1)

        def forward(self, x):
            return self.conv(x)

    class UNet(nn.Module):
        """Actor: 4 x G x G observation -> per-cell mean of a G x G density map + a learned log-std map (UnetActor.py:61-80)."""

        def __init__(self):
            super().__init__()
            self.inc = _Block(4, 64)
            self.down1 = _Down(64, 128)
            self.down2 = _Down(128, 256)
            self.up1 = _Up(384, 128)
            self.up2 = _Up(192, 64)
            self.out_mean = _Out(64, 1)
            self.log_std = nn.Parameter(torch.zeros((1, 1, map_size, map_size)))

        def forward(self, x):
            x1 = self.inc(x); x2 = self.down1(x1); x3 = self.down2(x2)
            mean = self.out_mean(self.up2(self.up1(x3, x2), x1))
            return mean[:, 0], self.log_std.expand_as(mean)[:, 0]      # [n, G, G] each (the reference squeezes: same for n > 1)

    class CNNCritic(nn.Module):
        """Critic: three 5x5 stride-2 convolutions, two linear layers (CNNCritic.py:7-49)."""

        def __init__(self):
            super().__init__()
            self.conv1 = _ortho(nn.Conv2d(4, 16, kernel_size=5, stride=2, padding=2))
            self.conv2 = _ortho(nn.Conv2d(16, 32, kernel_size=5, stride=2, padding=2))
            self.conv3 = _ortho(nn.Conv2d(32, 64, kernel_size=5, stride=2, padding=2))
            n = map_size
            for _ in range(3):
                n = (n - 1) // 2 + 1
            self.fc1 = _ortho(nn.Linear(64 * n * n, 100))                # 10816 inputs at G = 100
            self.fc2 = _ortho(nn.Linear(100, 1), std=1.0)

        def forward(self, x):
            x = F.relu(self.conv1(x)); x = F.relu(self.conv2(x)); x = F.relu(self.conv3(x))
            return self.fc2(F.relu(self.fc1(x.flatten(1))))

    return UNet, CNNCritic


# ------------------------------------------------------------------------------------------------------------------
# device-side transition buffers
# ------------------------------------------------------------------------------------------------------------------
class TransitionBuffers:
    """Per-charger transition lists of a batched roll-out, filled by the HIP kernels of csrc/wrsn_rollout.h.

    env          : VecWRSN (rendering on)
    capacity     : transitions kept per charger (further ones are counted in `count` and dropped)
    action_elems : size of the policy's raw output per decision: 3, or map_size**2 for density-map policies"""

    def __init__(self, env, capacity, action_elems):
        torch = env.torch
        if env.state is None:
            raise ValueError("TransitionBuffers needs a rendering VecWRSN (render=True)")
        self.env = env
        B, M, G, C, A = env.num_env, env.num_agent, env.map_size, int(capacity), int(action_elems)
        self.capacity, self.action_elems = C, A
        dev = env.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.pend_state = torch.zeros((B, M, 4, G, G), **f32)
        self.pend_action = torch.zeros((B, M, A), **f32)
        self.pend_logp = torch.zeros((B, M), **f32)
        self.pend_valid = torch.zeros((B, M), dtype=torch.uint8, device=dev)
        self.state = torch.zeros((M, C, 4, G, G), **f32)
        self.action = torch.zeros((M, C, A), **f32)
        self.next_state = torch.zeros((M, C, 4, G, G), **f32)
        self.reward = torch.zeros((M, C), **f32)
        self.logp = torch.zeros((M, C), **f32)
        self.now = torch.zeros((M, C), dtype=torch.float64, device=dev)
        self.env_index = torch.zeros((M, C), dtype=torch.int32, device=dev)
        self.count = torch.zeros((M,), dtype=torch.int32, device=dev)
        self._c = _lib.WrsnTransitionBuffers(C, A, *[t.data_ptr() for t in (
            self.pend_state, self.pend_action, self.pend_logp, self.pend_valid, self.state, self.action, self.next_state,
            self.reward, self.logp, self.now, self.env_index, self.count)])

    def clear(self, keep_pending=False):
        self.count.zero_()
        if not keep_pending:
            self.pend_valid.zero_()

    def record(self, agent_ids, actions, logp, states=None):
        """The chargers `agent_ids` [B] (< 0: none) are about to receive `actions` [B, action_elems] (IPPO.py:141-142)."""
        env, t = self.env, self.env.torch
        env._bind_stream()
        a = agent_ids.to(device=env.device, dtype=t.int32).contiguous()
        x = actions.to(device=env.device, dtype=t.float32).reshape(env.num_env, self.action_elems).contiguous()
        lp = logp.to(device=env.device, dtype=t.float32).reshape(env.num_env).contiguous()
        st = (env.state if states is None else states.to(device=env.device, dtype=t.float32)).contiguous()
        self._keep = (a, x, lp, st)                           # alive until the kernel has run
        env._h.rollout_record(self._c, a.data_ptr(), x.data_ptr(), lp.data_ptr(), st.data_ptr())

    def collect(self):
        """Append the transitions the request just returned by `env.step` completes (IPPO.py:144-155)."""
        env = self.env
        env._bind_stream()
        env._h.rollout_collect(self._c, **env._out_ptrs())

    def counts(self):
        """Transitions appended per charger so far (host list; synchronises)."""
        return [int(v) for v in self.count.cpu()]

    def stored(self):
        return [min(c, self.capacity) for c in self.counts()]


def select_batch(rewards, batch_size, rng=np.random):
    """The reference's batch selection (IPPO.py:193-200), index for index: the `batch_size // 2` rewards furthest from
    the mean, plus `batch_size - batch_size // 2` indices drawn without replacement from range(len - batch_size // 2)."""
    rewards = np.asarray(rewards)
    mean = np.mean(rewards)
    abs_diff = np.abs(rewards - mean)
    indices = np.argsort(abs_diff)
    selected_num = int(batch_size / 2.0)
    random_num = batch_size - selected_num
    return np.concatenate((indices[-selected_num:], rng.choice(len(rewards) - selected_num, size=random_num, replace=False)))


# ------------------------------------------------------------------------------------------------------------------
# trainer
# ------------------------------------------------------------------------------------------------------------------
DEFAULT_ARGS = dict(seed=0, lr=3.0e-4, gamma=0.99, clip=0.2, batch_size=512, n_updates_per_iteration=5, save_freq=5, gae=True,
                    norm_adv=True, minibatch_size=64, ent_coef=0.0, vf_coef=0.5, gae_lambda=0.95, max_grad_norm=0.5,
                    clip_vloss=True)                          # alg_args/ippo.yaml:4-18


class PPOLearner:
    """Networks, optimisers and the PPO arithmetic of the reference's IPPO (IPPO.py:17-117, 225-271) for `num_agent`
    chargers; knows nothing about environments.

    process_group : data-parallel training -- the flattened actor + critic gradients of the charger being updated are averaged
                  over the ranks with ONE all-reduce per minibatch (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).
                  Parameters and BatchNorm buffers are broadcast from rank 0 at construction."""

    def __init__(self, args, num_agent, map_size, device, model_path=None, infer_chunk=1024, process_group=None, inference_dtype=None, min_bucket=16):
        torch = _torch()
        self.torch = torch
        self.num_agent = int(num_agent)
        self.map_size = int(map_size)
        self.device = torch.device(device)
        a = dict(DEFAULT_ARGS); a.update(args or {})
        self.args = a
        for k in ("gamma", "clip", "batch_size", "minibatch_size", "n_updates_per_iteration", "save_freq", "gae", "clip_vloss", "ent_coef",
                  "vf_coef", "gae_lambda", "norm_adv", "max_grad_norm"):
            setattr(self, k, a[k])
        UNet, CNNCritic = build_networks(self.map_size)
        # channels-last weights and activations: the same float32 arithmetic, 3.6 x the NCHW convolution throughput of MIOpen on
        # MI355X (tools/diag_policy.py: 71 vs 20 TFLOP/s for the UNet forward); state_dict keys and shapes are unaffected
        self._cl = self.device.type == "cuda"
        mf = dict(memory_format=torch.channels_last) if self._cl else {}
        self.actors = [UNet().to(self.device).to(**mf) for _ in range(self.num_agent)]
        self.critics = [CNNCritic().to(self.device).to(**mf) for _ in range(self.num_agent)]
        # optional reduced-precision INFERENCE (roll-out actions and values only; the update stays float32): "bf16" doubles the
        # forward throughput again but the roll-out log-probabilities then differ from the float32 ones the update recomputes
        self.inference_dtype = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[inference_dtype]
        self.loggers = [{"i_so_far": 0, "t_so_far": 0, "ep_lifetime": [], "losses": [], "rewards": []} for _ in range(self.num_agent)]
        if model_path is not None:                            # IPPO.py:49-64
            for agent_folder in os.listdir(model_path):
                i = int(agent_folder); p = os.path.join(model_path, agent_folder)
                self.critics[i].load_state_dict(torch.load(os.path.join(p, "critic.pth"), map_location=self.device, weights_only=True))
                self.actors[i].load_state_dict(torch.load(os.path.join(p, "actor.pth"), map_location=self.device, weights_only=True))
        self.group = process_group
        dist = torch.distributed
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.world > 1:
            for net in self.actors + self.critics:
                for p in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(p.data, src=0, group=process_group)
        self.optimizers = [torch.optim.Adam(list(self.actors[i].parameters()) + list(self.critics[i].parameters()), lr=a["lr"])
                           for i in range(self.num_agent)]
        self.infer_chunk = int(infer_chunk)
        self.min_bucket = max(1, int(min_bucket))

    # -- policy ------------------------------------------------------------------------------------------------------
    def _bucket(self, r):
        """Rows a short inference chunk is padded to: `min_bucket` (16) doubled until it fits, at most `infer_chunk`.  MIOpen searches /
        compiles its convolution kernels per input shape (seconds for each new one), so the forward passes of a roll-out may only
        ever see this small set of batch shapes; padding to the power of two instead of the full chunk keeps a 20-row request
        from paying for (and taking its BatchNorm statistics from) hundreds of copies of itself."""
        b = self.min_bucket
        while b < r:
            b <<= 1
        return min(b, self.infer_chunk)

    def get_action(self, agent_id, states):
        """IPPO.py:95-105 for a batch: states [n,4,G,G] -> (action maps [n,G,G], summed log-prob [n]).

        The actor runs in training mode like the reference's (IPPO.py never calls `.eval()`), so BatchNorm normalises with the
        statistics of the rows in THIS forward pass: the sampled means and the stored log-probabilities depend on the batch
        composition (rows of the chunk + its padding), exactly as the reference's depend on its batch of one."""
        torch = self.torch
        outs, lps = [], []
        with torch.no_grad():
            for s in states.split(self.infer_chunk):
                r = s.shape[0]
                nb = self._bucket(r) if self.device.type == "cuda" else r
                if r < nb:                                    # a short chunk is filled up by repeating its own rows, whose outputs are dropped
                    s = s.index_select(0, torch.arange(nb, device=s.device) % r)
                mean, log_std = self._forward(self.actors[agent_id], s, inference=True)
                mean, log_std = mean[:r].float(), log_std[:r].float()
                dist = torch.distributions.Normal(mean, log_std.exp())
                act = dist.sample()
                outs.append(act); lps.append(dist.log_prob(act).sum((1, 2)))
        return torch.cat(outs), torch.cat(lps)

    def rollout_logp(self, agent_id, states, actions):
        """Log-probabilities of `actions` [n,G,G] under the current actor, evaluated over `states` in the SAME batch composition
        `get_action` uses (same chunks, same padding, same precision): with unchanged weights this reproduces the log-probabilities
        `get_action` returned for those rows.  `evaluate` over a different composition (a 64-row minibatch of the update) does not:
        BatchNorm in training mode makes the actor's output a function of the whole batch -- the reference has the same property
        (roll-out on batches of one, IPPO.py:95-105; update on minibatches, IPPO.py:236-241)."""
        torch = self.torch
        lps = []
        with torch.no_grad():
            for s, a in zip(states.split(self.infer_chunk), actions.split(self.infer_chunk)):
                r = s.shape[0]
                nb = self._bucket(r) if self.device.type == "cuda" else r
                if r < nb:
                    s = s.index_select(0, torch.arange(nb, device=s.device) % r)
                mean, log_std = self._forward(self.actors[agent_id], s, inference=True)
                dist = torch.distributions.Normal(mean[:r].float(), log_std[:r].float().exp())
                lps.append(dist.log_prob(a.float()).sum((1, 2)))
        return torch.cat(lps)

    def _forward(self, net, x, inference=False):
        torch = self.torch
        if self._cl:
            x = x.contiguous(memory_format=torch.channels_last)
        if inference and self.inference_dtype is not None:
            with torch.autocast(self.device.type, dtype=self.inference_dtype):
                return net(x)
        return net(x)

    def evaluate(self, agent_id, batch_states, batch_actions):  # IPPO.py:107-113
        torch = self.torch
        mean, log_std = self._forward(self.actors[agent_id], batch_states)
        dist = torch.distributions.Normal(mean, log_std.exp())
        return dist.log_prob(batch_actions).sum((1, 2)), dist.entropy().sum((1, 2))

    def get_value(self, agent_id, state):                     # IPPO.py:115-117
        return self._forward(self.critics[agent_id], state).sum(1)

    def _values(self, agent_id, states):
        torch = self.torch
        with torch.no_grad():
            return torch.cat([self._forward(self.critics[agent_id], s, inference=True).float().sum(1) for s in states.split(self.infer_chunk)])

    def cal_rt_adv(self, id, states, rewards, next_states, terminals):
        """IPPO.py:71-93.  `terminals` follows the reference's convention (the stored flags multiply the bootstrap term);
        every stored transition has terminal == False, so the recursion collapses to returns == rewards."""
        torch = self.torch
        with torch.no_grad():
            values = self._values(id, states)
            next_values = self._values(id, next_states)
            tm = terminals.to(rewards.dtype)
            if self.gae:
                advantages = torch.zeros_like(rewards)
                last = torch.zeros((), dtype=rewards.dtype, device=rewards.device)
                for t in reversed(range(len(rewards))):
                    delta = rewards[t] + self.gamma * next_values[t] * tm[t] - values[t]
                    last = delta + self.gamma * self.gae_lambda * tm[t] * last
                    advantages[t] = last
                returns = advantages + values
            else:
                # the reference's plain branch reads returns[t + 1] at t = len - 1 and raises IndexError on every non-empty
                # input (IPPO.py:86-90; alg_args/ippo.yaml ships gae: True): same behaviour here
                if len(rewards) > 0:
                    raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (len(rewards), len(rewards)))
                returns = torch.zeros_like(rewards); advantages = returns - values
        return returns, advantages, values

    # -- update ------------------------------------------------------------------------------------------------------
    def _allreduce_grads(self, id):
        """Data-parallel step: average the gradients of actor + critic of charger `id` over the ranks with
        one all-reduce of the flattened bucket."""
        if self.world <= 1:
            return
        torch = self.torch
        params = [p for p in list(self.actors[id].parameters()) + list(self.critics[id].parameters()) if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        torch.distributed.all_reduce(flat, group=self.group)
        flat /= self.world
        o = 0
        for p in params:
            n = p.numel(); p.grad.copy_(flat[o:o + n].view_as(p)); o += n

    def sync_buffers(self, id):
        """Data-parallel: BatchNorm running statistics are local to a rank (every rank normalises its own minibatches); average
        them over the ranks so that what is written to a checkpoint does not depend on which rank writes it."""
        if self.world <= 1:
            return
        torch = self.torch
        bufs = [b for b in list(self.actors[id].buffers()) + list(self.critics[id].buffers()) if b.dtype.is_floating_point]
        if not bufs:
            return
        flat = torch.cat([b.reshape(-1) for b in bufs])
        torch.distributed.all_reduce(flat, group=self.group)
        flat /= self.world
        o = 0
        for b in bufs:
            n = b.numel(); b.copy_(flat[o:o + n].view_as(b)); o += n

    def save_checkpoint(self, id, folder):
        """IPPO.py:296-309: `<folder>/{actor,critic}.pth`.  Under data-parallel training every rank calls this (the buffer average is
        a collective); rank 0 alone writes."""
        torch = self.torch
        self.sync_buffers(id)
        rank = torch.distributed.get_rank(self.group) if self.world > 1 else 0
        if rank == 0:
            os.makedirs(folder, exist_ok=True)
            torch.save(self.actors[id].state_dict(), os.path.join(folder, "actor.pth"))
            torch.save(self.critics[id].state_dict(), os.path.join(folder, "critic.pth"))

    def minibatch_loss(self, id, batch, mb):
        """The loss of IPPO.py:236-262 on the rows `mb` of `batch` for charger `id`: (loss, pg_loss, v_loss, entropy, approx_kl,
        clipfrac).  Forward passes run in training mode (BatchNorm statistics of this minibatch), like the reference's."""
        torch = self.torch
        newlogprob, entropy = self.evaluate(id, batch["states"][mb], batch["actions"][mb])
        newvalue = self.get_value(id, batch["states"][mb]).view(-1)
        logratio = newlogprob - batch["log_probs"][mb]
        ratio = logratio.exp()
        with torch.no_grad():
            approx_kl = ((ratio - 1) - logratio).mean()
            clipfrac = ((ratio - 1.0).abs() > self.clip).to(ratio.dtype).mean().item()
        adv = batch["advantages"][mb]
        if self.norm_adv:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        pg_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - self.clip, 1 + self.clip)).mean()
        if self.clip_vloss:
            v_un = (newvalue - batch["returns"][mb]) ** 2
            v_cl = batch["values"][mb] + torch.clamp(newvalue - batch["values"][mb], -self.clip, self.clip)
            v_loss = 0.5 * torch.max(v_un, (v_cl - batch["returns"][mb]) ** 2).mean()
        else:
            v_loss = 0.5 * ((newvalue - batch["returns"][mb]) ** 2).mean()
        entropy_loss = entropy.mean()
        loss = pg_loss - self.ent_coef * entropy_loss + v_loss * self.vf_coef
        return loss, pg_loss, v_loss, entropy_loss, approx_kl, clipfrac

    def apply_gradients(self, id):
        """IPPO.py:264-268 after `loss.backward()`: (data-parallel: average the gradients over the ranks,) clip the actor's and the
        critic's gradient norms separately, Adam step."""
        nn = self.torch.nn
        self._allreduce_grads(id)
        nn.utils.clip_grad_norm_(self.actors[id].parameters(), self.max_grad_norm)
        nn.utils.clip_grad_norm_(self.critics[id].parameters(), self.max_grad_norm)
        self.optimizers[id].step()

    def update(self, id, batch, shuffle=np.random.shuffle):
        """IPPO.py:229-271 for charger `id`; returns the last minibatch's (pg_loss, v_loss, entropy, approx_kl, clipfrac).
        `approx_kl` / `clipfrac` of the FIRST minibatch after a roll-out are not small, here as in the reference: the stored
        log-probabilities were computed with the BatchNorm statistics of the roll-out batch (`get_action`; the reference: a batch of
        one), the update recomputes them over a minibatch -- see `rollout_logp`."""
        torch = self.torch
        b_inds = np.arange(self.batch_size)
        clipfracs = []
        stats = None
        for _ in range(self.n_updates_per_iteration):
            shuffle(b_inds)
            for start in range(0, self.batch_size, self.minibatch_size):
                mb = torch.as_tensor(b_inds[start:start + self.minibatch_size], device=batch["states"].device, dtype=torch.long)
                loss, pg_loss, v_loss, entropy_loss, approx_kl, clipfrac = self.minibatch_loss(id, batch, mb)
                clipfracs.append(clipfrac)
                self.optimizers[id].zero_grad()
                loss.backward()
                self.apply_gradients(id)
                self.loggers[id]["losses"].append(float(loss.detach()))
                stats = (float(pg_loss.detach()), float(v_loss.detach()), float(entropy_loss.detach()), float(approx_kl), float(np.mean(clipfracs)))
        return stats


class BatchedIPPO(PPOLearner):
    """`IPPO(args, env, device, model_path=None)` of the reference (IPPO.py:17-69) over a `VecWRSN` with density-map actions
    (`density_map=True` environments of runner/IPPO.py:19-21): the actor's G x G output is the action and is turned into the
    3-vector on the device (`VecWRSN.density_to_action`)."""

    def __init__(self, args, env, device=None, model_path=None, capacity=None, infer_chunk=1024, process_group=None, log=None, inference_dtype=None,
                 min_bucket=16):
        super().__init__(args, env.num_agent, env.map_size, device if device is not None else env.device, model_path, infer_chunk, process_group,
                         inference_dtype, min_bucket)
        self.env = env
        self.buffers = TransitionBuffers(env, capacity or 2 * self.batch_size, env.map_size * env.map_size)
        self.timers = {"env_s": 0.0, "policy_s": 0.0, "glue_s": 0.0, "train_s": 0.0, "launches": 0, "requests": 0}
        self.log = log
        self._req = None

    # -- roll-out ----------------------------------------------------------------------------------------------------
    def _sync_time(self):
        self.torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        return time.perf_counter()

    def step_batch(self):
        """One launch of the batched roll-out: act for every environment that carries a request, step, collect."""
        torch, env = self.torch, self.env
        r = self._req
        ids = r["agent_id"]
        G = env.map_size
        t0 = self._sync_time()
        maps = torch.zeros((env.num_env, G * G), dtype=torch.float32, device=env.device)
        logp = torch.zeros((env.num_env,), dtype=torch.float32, device=env.device)
        for a in range(self.num_agent):
            rows = torch.nonzero(ids == a).flatten()
            if rows.numel() == 0:
                continue
            act, lp = self.get_action(a, r["state"].index_select(0, rows))
            maps.index_copy_(0, rows, act.reshape(rows.numel(), G * G).float()); logp.index_copy_(0, rows, lp.float())
        t1 = self._sync_time()
        self.buffers.record(ids, maps, logp)
        act3 = env.density_to_action(ids, maps.view(env.num_env, G, G).double())
        t2 = self._sync_time()
        r = env.step(ids.clone(), act3)
        t3 = self._sync_time()
        self.buffers.collect()
        t4 = self._sync_time()
        tm = self.timers
        tm["policy_s"] += t1 - t0; tm["glue_s"] += (t2 - t1) + (t4 - t3); tm["env_s"] += t3 - t2; tm["launches"] += 1
        tm["requests"] += int((ids >= 0).sum())
        self.last_ids, self.last_action3 = ids, act3          # what this launch handed to the environments (tests / logging)
        self._req = r
        return r

    def roll_out(self, max_launches=100000, fresh_episodes=False):
        """IPPO.py:119-210 over the batch: launches until every charger has `batch_size` transitions, then the reference's
        per-charger batch selection.  Environments restart by auto-reset, so one roll-out spans many episodes.

        The environments live ACROSS roll-outs: the first call resets them, every later call goes on from the requests the
        previous one ended with (pending actions included -- their stored log-probabilities come from the policy that chose
        them, i.e. the one before the last update).  With a large batch the per-charger quota is reached within a few launches;
        restarting every roll-out from `reset()` would then only ever store the first decision after the warm-up snapshot (all
        chargers at the base station, no node death, no terminal return), which is not what the reference's roll_out collects: it
        runs every episode to its terminal (IPPO.py:130-190).  `fresh_episodes=True` restores the restart-per-roll-out behaviour."""
        torch, env = self.torch, self.env
        if not env.auto_reset:
            raise ValueError("BatchedIPPO needs VecWRSN(auto_reset=True)")
        if fresh_episodes or self._req is None:
            self.buffers.clear()
            self._req = env.reset()
        else:
            self.buffers.clear(keep_pending=True)
        for _ in range(max_launches):
            self.step_batch()
            if min(self.buffers.counts()) >= self.batch_size:
                break
        short = [a for a, n in enumerate(self.buffers.stored()) if n < self.batch_size]
        if short:
            raise RuntimeError("roll_out stopped after %d launches with %s transitions per charger, fewer than batch_size %d (buffer capacity %d): "
                               "raise max_launches / capacity or lower batch_size" % (max_launches, self.buffers.stored(), self.batch_size, self.buffers.capacity))
        out = []
        for a in range(self.num_agent):
            n = self.buffers.stored()[a]
            rewards = self.buffers.reward[a, :n]
            idx_np = select_batch(rewards.cpu().numpy(), self.batch_size)
            idx = torch.as_tensor(idx_np, device=env.device, dtype=torch.long)
            states = self.buffers.state[a].index_select(0, idx); nxt = self.buffers.next_state[a].index_select(0, idx)
            rew = rewards.index_select(0, idx)
            # the reference runs cal_rt_adv per episode over the transitions of that episode (IPPO.py:171); with terminals all
            # False neither returns nor advantages couple two transitions, so one call over the selected batch gives the same values
            returns, adv, values = self.cal_rt_adv(a, states, rew, nxt, torch.zeros_like(rew))
            out.append(dict(states=states, actions=self.buffers.action[a].index_select(0, idx).view(-1, env.map_size, env.map_size),
                            log_probs=self.buffers.logp[a].index_select(0, idx), rewards=rew, next_states=nxt, advantages=adv, returns=returns,
                            values=values))
            self.loggers[a]["rewards"].append(float(rew.mean()))
        return out

    def train(self, trained_iterations, save_folder=None):
        """IPPO.py:212-311: roll out, update every charger, log, checkpoint every `save_freq` iterations."""
        torch = self.torch
        start = time.time(); i_so_far = 0; rows = []
        while i_so_far <= trained_iterations:
            batches = self.roll_out()
            i_so_far += 1
            t0 = self._sync_time()
            for id in range(self.num_agent):
                lg = self.loggers[id]
                lg["t_so_far"] += self.batch_size; lg["i_so_far"] += 1
                st = self.update(id, batches[id])
                y_pred, y_true = batches[id]["values"].cpu().numpy(), batches[id]["returns"].cpu().numpy()
                var_y = np.var(y_true)
                ev = float("nan") if var_y == 0 else 1 - np.var(y_true - y_pred) / var_y
                row = dict(iteration=lg["i_so_far"], timesteps=lg["t_so_far"], agent=id, policy_loss=st[0], value_loss=st[1], entropy=st[2],
                           approx_kl=st[3], clipfrac=st[4], explained_variance=float(ev), mean_reward=lg["rewards"][-1],
                           sps=i_so_far / (time.time() - start))
                rows.append(row)
                if self.log:
                    self.log(row)
                if save_folder is not None and lg["i_so_far"] % self.save_freq == 0:      # IPPO.py:296-309 layout: <iter>/<agent>/{actor,critic}.pth
                    self.save_checkpoint(id, os.path.join(save_folder, str(lg["i_so_far"]), str(id)))
            self.timers["train_s"] += self._sync_time() - t0
        return rows
