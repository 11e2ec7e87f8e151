"""PPO inner loop on the device: observation ring, actor/critic MLP (fp32 MFMA), GAE,
landing reward and the rollout collector.

Reference (relative to /root/reference):
  environment/controller/dl_auxiliary.py:15-32   dl_in_gen
  environment/controller/model.py:20-85          ActorCritic
  environment/controller/ppo.py:125-141,211-269  get_advantages, worker rollouts
  visual_landing/ppo_trainer.py:133-148          get_advantages (ddof=1)
  visual_landing/rl_reward_fuction.py:3-102      visual_reward
"""
import ctypes as C
import os
import itertools
import weakref

import numpy as np
import torch

from . import _lib as L
from .quadrotor_env import BatchedQuad, quad


def _stream(dev):
    """The caller's current stream on `dev` as a C pointer; `dev` becomes the thread's current device (a kernel launched
    into a stream of another device than the current one fails)."""
    dev = torch.device(dev)
    if dev.index is not None and torch.cuda.current_device() != dev.index:
        torch.cuda.set_device(dev)
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# One version cell per weight STORAGE, shared by every ActorCriticWeights that wraps the same tensors:
# `ActorCriticWeights(w.actor, None)` is such an alias (dev() hands fp32 device tensors back as they are), and an env that
# holds the alias must see the learner's touch() of the object it trains.  Ownership: the wrappers hold their cells strongly;
# the registry (device pointer of the first layer -> cell) holds them weakly, so an entry goes when the last wrapper of a
# storage goes, and a cell is registered under ONE address -- its home -- at a time (re-homing drops the old key: a later,
# unrelated tensor that the caching allocator places at a freed address starts a cell of its own).  Version numbers come
# from one process-wide counter, so a number is never handed out twice.
class _VersionCell:
    __slots__ = ("v", "home", "__weakref__")

    def __init__(self, home):
        self.v = next(_NEXT_VERSION)
        self.home = home          # data_ptr() of the first layer where the trained values live now


_NEXT_VERSION = itertools.count(1)
_VERSION_CELLS = weakref.WeakValueDictionary()


def _version_cell(t_in, t_dev):
    p = t_dev.data_ptr()
    alias = isinstance(t_in, torch.Tensor) and t_in.device == t_dev.device and t_in.data_ptr() == p
    cell = _VERSION_CELLS.get(p) if alias else None
    if cell is None:   # our own copy of the caller's numbers (or a storage nobody registered): a new cell
        cell = _VersionCell(p)
        _VERSION_CELLS[p] = cell
    return cell


class ActorCriticWeights:
    """float32 device copies of model.py's actor / critic Linear layers, torch layout
    ([out, in] row-major), read directly by the MFMA kernel."""

    def __init__(self, actor, critic, device="cuda", std=0.1):
        self.device = torch.device(device)
        def dev(t):
            t = t.detach() if isinstance(t, torch.Tensor) else torch.from_numpy(np.asarray(t))
            return t.to(device=self.device, dtype=torch.float32).contiguous()
        self.actor = [dev(t) for t in actor]
        self.critic = None if critic is None else [dev(t) for t in critic]
        self.std = float(std)
        self.hidden = self.actor[0].shape[0]
        self.state_dim = self.actor[0].shape[1]
        # bumped whenever the weight VALUES change in place (PpoLearner does; call touch() after writing into the tensors
        # yourself): an env re-builds its prepared image of the weights (qenv_mlp_prepare) when it sees a new version.  The
        # cells belong to the storage, not to this wrapper (see _version_cell): aliases of the same tensors share them.
        self._cells = {"actor": _version_cell(actor[0], self.actor[0])}
        if self.critic is not None:
            self._cells["critic"] = _version_cell(critic[0], self.critic[0])

    def version_of(self, head):
        """The version of `head`'s storage.  A wrapper whose tensors are no longer where the cell's values live -- an alias
        made BEFORE a PpoLearner re-homed the weights into its flat parameter buffer -- holds stale copies and raises."""
        t = self.head(head)[0]
        cell = self._cells[head]
        if cell.home != t.data_ptr():
            raise RuntimeError(f"ActorCriticWeights: this wrapper's {head} tensors are not the trained ones any more (a PpoLearner "
                               "moved the weights into its flat parameter buffer after this alias was made); make the alias "
                               "(actor_only()) after constructing the learner")
        return cell.v

    def _rehome(self):
        """The wrapper's tensors were replaced by views of another buffer holding the same values (PpoLearner's flat parameter
        buffer): the cells follow them -- the old address is released -- and get a new version."""
        for head in ("actor", "critic"):
            if head == "critic" and self.critic is None:
                continue
            cell, p = self._cells[head], self.head(head)[0].data_ptr()
            if _VERSION_CELLS.get(cell.home) is cell:
                del _VERSION_CELLS[cell.home]
            cell.home = p
            _VERSION_CELLS[p] = cell
            cell.v = next(_NEXT_VERSION)

    @property
    def version(self):
        return self.version_of("actor")

    def touch(self):
        """Tell the envs that the weight values changed in place (their prepared images are rebuilt at the next forward);
        every wrapper of the same tensors sees it."""
        for head in ("actor", "critic"):
            if head == "critic" and self.critic is None:
                continue
            self.version_of(head)   # (a stale alias raises here rather than announce values it does not hold)
            self._cells[head].v = next(_NEXT_VERSION)

    @classmethod
    def from_state_dict(cls, sd, device="cuda"):
        """`sd`: torch state_dict of model.py's ActorCritic (keys actor.{0,2,4}.{weight,bias}, ...)
        or the same keys with '_' (re-exported .npz)."""
        def get(k):
            v = sd[k] if k in sd else sd[k.replace(".", "_")]
            return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        actor = [get(f"actor.{l}.{p}") for l in (0, 2, 4) for p in ("weight", "bias")]
        has_critic = ("critic.0.weight" in sd) or ("critic_0_weight" in sd)
        critic = [get(f"critic.{l}.{p}") for l in (0, 2, 4) for p in ("weight", "bias")] if has_critic else None
        std = float(get("std")) if ("std" in sd) else 0.1
        return cls(actor, critic, device, std)

    @classmethod
    def random(cls, hidden, state_dim=75, action_dim=4, device="cuda", seed=0):
        """nn.Linear default init (uniform +-1/sqrt(fan_in)) from a numpy stream."""
        rng = np.random.RandomState(seed)

        def lin(o, i):
            b = 1.0 / np.sqrt(i)
            return rng.uniform(-b, b, (o, i)), rng.uniform(-b, b, o)
        def mlp(out):
            W1, b1 = lin(hidden, state_dim); W2, b2 = lin(hidden, hidden); W3, b3 = lin(out, hidden)
            return [W1, b1, W2, b2, W3, b3]
        return cls(mlp(action_dim), mlp(1), device)

    def actor_only(self):
        """A wrapper of the SAME actor tensors without the critic (what an evaluation env needs).  It shares the actor's
        version cell, so an env holding it rebuilds its prepared image after the learner's touch().  Make it AFTER
        constructing a PpoLearner on these weights: the learner re-homes the tensors into its flat parameter buffer, and a
        wrapper of the tensors from before that move keeps the old, no longer trained, copies."""
        w = ActorCriticWeights(self.actor, None, self.device, self.std)
        return w

    def head(self, name):
        if name == "actor":
            return self.actor
        if self.critic is None:
            raise ValueError("no critic weights loaded")
        return self.critic


class dl_in_gen:
    """dl_auxiliary.py:15-32 for one `quad`: the T-slice window lives in the env's device
    ring; dl_input() pushes slices and returns the window in the reference's time order."""

    def __init__(self, T, state_size, action_size, env=None):
        if env is None:
            raise ValueError("dl_in_gen(T, state_size, action_size, env=<quad>): the window is kept on the device of "
                             "the env it observes")
        self._b = env._b if isinstance(env, quad) else env
        if self._b.T != T:
            raise ValueError(f"window length T={T} must equal the env's T={self._b.T}")
        self.hist_size = (state_size - 3) + action_size + 1
        self.deep_learning_in_size = self.hist_size * T
        self.T = T
        self._pushes = 0
        self._pending_reset = True
        self.deep_learning_input = np.zeros(self.deep_learning_in_size, dtype=np.float32)

    def reset(self):
        self._pending_reset = True
        self._pushes = 0
        self.deep_learning_input = np.zeros(self.deep_learning_in_size, dtype=np.float32)

    def dl_input(self, states, actions):
        b = self._b
        for s, a in zip(np.atleast_2d(states), np.atleast_2d(actions)):
            b.window_push(torch.from_numpy(np.asarray(s, dtype=np.float64).reshape(14, 1)),
                          torch.from_numpy(np.asarray(a, dtype=np.float64).reshape(4, 1)), self._pending_reset)
            self._pending_reset = False
            self._pushes += 1
        ring = b.window[:, 0].cpu().numpy().reshape(self.T, self.hist_size)
        head = self._pushes % self.T  # oldest slice
        self.deep_learning_input = np.concatenate([ring[(head + j) % self.T] for j in range(self.T)]).astype(np.float32)
        return self.deep_learning_input


def gae(values, masks, rewards, ddof=0, group=None, normalize=True, deterministic=False, return_stats=False, valid=None):
    """PPO.get_advantages (ppo.py:125-141 ddof=0; ppo_trainer.py:133-148 ddof=1) on a
    time-major rollout: values [n+1, N], masks [n, N], rewards [n, N] device tensors.
    Returns (returns [n, N], normalised advantages [n, N]).  With `group`
    (torch.distributed) the normalisation moments are all-reduced so every shard uses the
    global mean/std, exactly like a single learner.  deterministic: the moments are summed in a fixed order instead
    of with double-precision atomics.  return_stats (with normalize=False): also return the three float64 moment
    sums [count, sum, sum of squares] of the raw advantages, for a caller that combines several shards before
    `adv_normalize`.  valid [n, N] (in-band warm-up rollouts): rows with 0 are a restarted env's warm-up steps -- their
    advantage is 0, before and after the normalisation, and the moments (count included) leave them out."""
    n, N = rewards.shape
    assert values.shape == (n + 1, N) and masks.shape == (n, N)
    dt = L.F64 if values.dtype == torch.float64 else L.F32
    if values.dtype not in (torch.float32, torch.float64):
        raise ValueError("gae: values must be float32 or float64")
    values, masks, rewards = values.contiguous(), masks.to(values.dtype).contiguous(), rewards.to(values.dtype).contiguous()
    ret = torch.empty_like(rewards)
    adv = torch.empty_like(rewards)
    stats = torch.zeros(3, dtype=torch.float64, device=values.device)
    stream = _stream(values.device)
    if valid is not None:
        assert valid.shape == (n, N)
        valid = valid.to(values.dtype).contiguous()
    ws = torch.empty(int(L.lib.qenv_gae_workspace_bytes(N)), dtype=torch.uint8, device=values.device) if deterministic else None
    if valid is not None:
        L.check(L.lib.qenv_gae_masked(dt, N, n, L.ptr(values), L.ptr(masks), L.ptr(rewards), L.ptr(valid), L.ptr(ret), L.ptr(adv),
                                      L.ptr(stats), L.ptr(ws), stream), "qenv_gae_masked")
    elif deterministic:  # moments summed in a fixed order (run-to-run identical normalised advantages)
        L.check(L.lib.qenv_gae_deterministic(dt, N, n, L.ptr(values), L.ptr(masks), L.ptr(rewards), L.ptr(ret), L.ptr(adv),
                                             L.ptr(stats), L.ptr(ws), stream), "qenv_gae_deterministic")
    else:
        L.check(L.lib.qenv_gae(dt, N, n, L.ptr(values), L.ptr(masks), L.ptr(rewards), L.ptr(ret), L.ptr(adv),
                               L.ptr(stats), stream), "qenv_gae")
    if not normalize:
        return (ret, adv, stats) if return_stats else (ret, adv)
    if group is not False and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(stats, group=group)
    adv_normalize(adv, stats, ddof, valid=valid)
    return (ret, adv, stats) if return_stats else (ret, adv)


def adv_normalize(adv, stats, ddof=0, valid=None):
    """(adv - mean) / (std + 1e-10) in place from the float64 moment sums `gae(..., normalize=False, return_stats=True)`
    returned (summed over shards / all-reduced over ranks by the caller): ppo.py:139, ppo_trainer.py:146."""
    if adv.dtype not in (torch.float32, torch.float64) or not adv.is_contiguous() or stats.dtype != torch.float64 or stats.numel() < 3:
        raise ValueError("adv_normalize: adv must be a contiguous float32 / float64 tensor, stats three float64 sums")
    dt = L.F64 if adv.dtype == torch.float64 else L.F32
    if valid is not None:
        if valid.shape != adv.shape:
            raise ValueError("adv_normalize: valid must have adv's shape")
        valid = valid.to(adv.dtype).contiguous()
    L.check(L.lib.qenv_adv_normalize_masked(dt, adv.numel(), L.ptr(adv), L.ptr(valid), L.ptr(stats), int(ddof), _stream(adv.device)),
            "qenv_adv_normalize")
    return adv


def dense_tail(x, weights, final_tanh=True, out=None, rows=False):
    """Dense tail of visual_landing/rl_nn_model.py (:58-66): Linear(K,H) tanh Linear(H,H) tanh Linear(H,n_out)
    [tanh] on feature-major float32 x [K, N] -- or, rows=True, on x [N, K], the layout the reference's own model feeds this
    tail (its conv features flattened to [batch, 800]): no transposed copy to make (hidden 64, K % 8 == 0, >= 65 536
    samples; the same speed), bit-identical to the feature-major call on x.t().  weights = [W1, b1, W2, b2, W3, b3]
    float32 device tensors in torch layout.  Returns [n_out, N]."""
    W1, b1, W2, b2, W3, b3 = [w.to(device=x.device, dtype=torch.float32).contiguous() for w in weights]
    K, N = (x.shape[1], x.shape[0]) if rows else x.shape
    assert W1.shape[1] == K and x.dtype == torch.float32
    x = x.contiguous()
    if rows:
        out = torch.empty(W3.shape[0], N, device=x.device, dtype=torch.float32) if out is None else out
        if tuple(out.shape) != (W3.shape[0], N) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"dense_tail: out must be a contiguous float32 [{W3.shape[0]}, {N}] tensor on {x.device}")
        L.check(L.lib.qenv_dense_tail_rows(N, K, W1.shape[0], W3.shape[0], int(bool(final_tanh)), L.ptr(x), L.ptr(W1), L.ptr(b1),
                                           L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3), L.ptr(out), _stream(x.device)), "qenv_dense_tail_rows")
        return out
    out = torch.empty(W3.shape[0], N, device=x.device, dtype=torch.float32) if out is None else out
    if tuple(out.shape) != (W3.shape[0], N) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"dense_tail: out must be a contiguous float32 [{W3.shape[0]}, {N}] tensor on {x.device}")
    stream = _stream(x.device)
    L.check(L.lib.qenv_dense_tail(N, K, W1.shape[0], W3.shape[0], int(bool(final_tanh)), L.ptr(x), L.ptr(W1), L.ptr(b1),
                                  L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3), L.ptr(out), stream), "qenv_dense_tail")
    return out


def ppo_loss(mean, value, actions, old_logprob, adv, returns, std, eps_clip=0.2, c_value=0.5, c_entropy=0.006,
             with_grads=False):
    """Loss of PPO.update (ppo.py:185-203) for a batch: mean/actions/old_logprob [4, B], value/adv/returns [B].
    Returns a dict of device scalars {loss, actor, critic, entropy} (loss = loss.mean() of the reference) and,
    with_grads, d loss / d mean [4, B] and d loss / d value [B]."""
    B = adv.numel()
    dt = L.F64 if mean.dtype == torch.float64 else L.F32
    c = lambda t: t.to(mean.dtype).contiguous()
    mean, value, actions, old_logprob, adv, returns = map(c, (mean, value, actions, old_logprob, adv, returns))
    sums = torch.zeros(4, dtype=torch.float64, device=mean.device)
    gm = torch.empty_like(mean) if with_grads else None
    gv = torch.empty_like(value) if with_grads else None
    stream = _stream(mean.device)
    L.check(L.lib.qenv_ppo_loss(dt, B, L.ptr(mean), L.ptr(value), L.ptr(actions), L.ptr(old_logprob), L.ptr(adv),
                                L.ptr(returns), float(std), float(eps_clip), float(c_value), L.ptr(sums), L.ptr(gm),
                                L.ptr(gv), stream), "qenv_ppo_loss")
    actor = sums[1] / sums[0]
    critic = c_value * sums[2] / sums[0]
    entropy = -c_entropy * sums[3] / sums[0]
    out = {"loss": actor + critic + entropy, "actor": actor, "critic": critic, "entropy": entropy}
    if with_grads:
        out["grad_mean"], out["grad_value"] = gm, gv
    return out


class PpoLearner:
    """PPO.update of environment/controller/ppo.py:143-209 on the device for the 75-N-N-4 actor / 75-N-N-1
    critic of model.py: K_epochs full-batch passes of { forward, clipped-surrogate loss, backward (fp32 MFMA),
    Adam } over a `PolicyRollout` collected with store_obs=True.  FIXED_STD as in the reference (:301).

    Gradients of the weights are fp32-atomic sums over the batch (order not deterministic to the last bit) unless
    deterministic=True.  The learner moves the parameters into one flat buffer: `weights.actor[i]` / `.critic[i]`
    become views of it (same values), so read the tensors through the `ActorCriticWeights` object after constructing
    the learner -- a tensor taken from the lists before that no longer follows the updates."""

    def __init__(self, weights: ActorCriticWeights, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, eps_clip=0.2, c_value=0.5,
                 c_entropy=0.006, K_epochs=10, group=None, deterministic=False, learn_std=False):
        """group: torch.distributed group of the data-parallel learners (None = the default group if one is
        initialised, False = never reduce).  Each rank holds an equal shard of the batch; gradients are averaged
        with one bucketed all-reduce per epoch, so every rank applies the same Adam step.
        learn_std: the action std is a trained parameter (model.py:45-48 with fixed_std=False; the reference's default is
        FIXED_STD, ppo.py:301): d loss / d std = mean_s d actor_s / d std - 4 c_entropy / std joins the gradient bucket
        as its last element, Adam updates it with the weights, `weights.std` follows after every step."""
        assert weights.critic is not None
        self.learn_std = bool(learn_std)
        self.w, self.lr, self.betas, self.eps = weights, lr, betas, eps
        self.eps_clip, self.c_value, self.c_entropy, self.K_epochs = eps_clip, c_value, c_entropy, K_epochs
        self.step = 0
        self.group = group
        self.fused_loss = weights.hidden in (16, 32, 64, 128)  # False: forward, qenv_ppo_loss and backward as separate launches
        # hidden 128, 75 inputs: the head's backward pass and weight gradients as ONE launch (qenv_mlp_grad_ppo, round 5);
        # False / QENV_GRAD_ONE_LAUNCH=0: the backward kernel + three weight-gradient launches of rounds 1-4 (A/B runs)
        self.one_launch = os.environ.get("QENV_GRAD_ONE_LAUNCH", "1") != "0"
        # True: weight / bias gradients summed in a fixed order (per-workgroup slots + an ordered second pass) instead of
        # fp32 atomics -- with gae(deterministic=True) a training run is bit-reproducible; ~3 % slower at 8.4 M samples
        self.deterministic = bool(deterministic)
        self._wg_ws = {}
        from .distributed import GradBucket
        keys = [(head, k) for head in ("actor", "critic") for k in range(6)]
        # parameters, gradients and both Adam moments each live in ONE flat buffer (the weight tensors become views of
        # it): one zeroing launch and one Adam launch per step instead of twelve of each, one all-reduce bucket
        params = [weights.head(h)[k] for h, k in keys]
        extra = [torch.tensor([float(weights.std)], device=params[0].device, dtype=torch.float32)] if self.learn_std else []
        self.flat_p = torch.cat([p.reshape(-1) for p in params] + extra).contiguous()
        off = 0
        for (h, k), p in zip(keys, params):
            weights.head(h)[k] = self.flat_p[off:off + p.numel()].view_as(p)
            off += p.numel()
        weights._rehome()   # the tensors moved into the flat buffer: the version cells follow them to their new home
        if self.learn_std and not self.fused_loss:
            raise ValueError("learn_std needs the fused-loss backward (hidden 16, 32, 64 or 128)")
        self.std_param = self.flat_p[-1:] if self.learn_std else None
        # every .grad is a slice of one flat buffer (the std's gradient last)
        self.bucket = GradBucket([weights.head(h)[k] for h, k in keys] + ([self.std_param] if self.learn_std else []))
        self.flat_m, self.flat_v = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        self._state = {}
        off = 0
        for key, g in zip(keys, self.bucket.views):
            n = g.numel()
            self._state[key] = (g, self.flat_m[off:off + n].view_as(g), self.flat_v[off:off + n].view_as(g))  # grad, m, v
            off += n
        self._buf = None

    def _buffers(self, H, n, N, dev):
        key = (H, n, N)
        if self._buf is None or self._buf[0] != key:
            mk = lambda r: torch.empty(n, r, N, device=dev, dtype=torch.float32)
            self._buf = (key, dict(h1=mk(H), h2=mk(H), dz2=mk(H), dz1=mk(H), dz3=mk(4), mean=mk(4), value=mk(1)))
        return self._buf[1]

    def _world(self):
        """Ranks that share this learner's gradients (1: no group, group=False, or no initialised process group)."""
        import torch.distributed as dist
        if self.group is False or not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _global_count(self, n_local, dev):
        """Sum of a per-rank sample count over the learner's group (the denominator of a loss that is a mean over ALL the
        ranks' valid samples)."""
        if self._world() == 1:
            return int(n_local)
        import torch.distributed as dist
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def gradients(self, obs, actions, old_logprob, adv, returns, valid=None, n_valid=None, n_valid_global=None):
        """One forward + loss + backward over the batch; fills the .grad buffers.  obs [n, 75, N] float32,
        actions / old_logprob [n, 4, N], adv / returns [n, N].  Returns the loss dict of `ppo_loss`.
        valid [n, N] (in-band warm-up rollouts): rows of weight 0 take no part in the loss; n_valid = their count on THIS
        learner (computed here with one device-to-host read if not given).  Data-parallel learners hold DIFFERENT numbers
        of valid rows, so the loss is normalised by n_valid_global (their sum over the group: one 8-byte all-reduce if not
        given) and the gradients are SUMMED over the ranks -- the single learner's sum / count, not a mean of per-rank
        means."""
        n, K, N = obs.shape
        dev, w = obs.device, self.w
        H = w.hidden
        # (the one-launch gradient path keeps no activations: no [n, H, N] buffers -- 17 GB at config 5's horizon)
        b = None if (self.fused_loss and self.one_launch and H == 128 and K == 75) else self._buffers(H, n, N, dev)
        stream = _stream(dev)
        f32 = lambda t: t.to(torch.float32).contiguous()
        # (the one-launch gradient kernel reads a rollout's observations where they were stored: a [n, 75, N] VIEW of the slice
        # array -- chunk stride 15 N, rows contiguous -- needs no dense copy)
        in_place = (self.fused_loss and self.one_launch and H == 128 and K == 75 and obs.dtype == torch.float32
                    and obs.stride(2) == 1 and obs.stride(1) == N and (n == 1 or obs.stride(0) >= N))
        actions, old_logprob, adv, returns = map(f32, (actions, old_logprob, adv, returns))
        if not in_place:
            obs = f32(obs)
        self.bucket.zero_()
        if valid is not None:
            if not self.fused_loss:
                raise ValueError("valid rows need the fused-loss backward (hidden 16, 32, 64 or 128)")
            valid = f32(valid)
            if n_valid is None:
                n_valid = int(valid.sum().item())
            if n_valid_global is None:
                n_valid_global = self._global_count(n_valid, dev)
        if self.fused_loss:
            return self._gradients_fused(obs, actions, old_logprob, adv, returns, b, valid, n_valid, n_valid_global)
        # forward of both heads on the stored observations (the dense kernel takes feature-major chunks)
        for head, out, ft in (("actor", b["mean"], True), ("critic", b["value"], False)):
            W = w.head(head)  # all n stored steps in one launch
            L.check(L.lib.qenv_dense_tail_chunks(N, n, K, H, W[4].shape[0], int(ft), L.ptr(obs), *[L.ptr(x) for x in W],
                                                 L.ptr(out), stream), "qenv_dense_tail_chunks")
        # loss + seed gradients over the whole batch: view [n, r, N] as r rows of n*N samples
        flat = lambda t, r: t.permute(1, 0, 2).reshape(r, n * N).contiguous()
        loss = ppo_loss(flat(b["mean"], 4), b["value"].reshape(n * N), flat(actions, 4), flat(old_logprob, 4),
                        adv.reshape(n * N), returns.reshape(n * N), w.std, self.eps_clip, self.c_value, self.c_entropy,
                        with_grads=True)
        seed_a = loss["grad_mean"].reshape(4, n, N).permute(1, 0, 2).contiguous()
        seed_c = loss["grad_value"].reshape(n, 1, N).contiguous()
        for head, seed, ft in (("actor", seed_a, True), ("critic", seed_c, False)):
            W = w.head(head)
            n_out = W[4].shape[0]
            L.check(L.lib.qenv_mlp_backward(N, n, K, H, n_out, int(ft), L.ptr(obs), L.ptr(seed), *[L.ptr(x) for x in W],
                                            L.ptr(b["h1"]), L.ptr(b["h2"]), L.ptr(b["dz2"]), L.ptr(b["dz1"]), L.ptr(b["dz3"]),
                                            stream), "qenv_mlp_backward")
            g = [self._state[(head, k)][0] for k in range(6)]
            dz3 = b["dz3"] if n_out == 4 else b["dz3"][:, :n_out].contiguous()
            for A, rows_a, Bm, rows_b, gw, gb in ((dz3, n_out, b["h2"], H, g[4], g[5]), (b["dz2"], H, b["h1"], H, g[2], g[3]),
                                                 (b["dz1"], H, obs, K, g[0], g[1])):
                self._weight_grad(N, n, rows_a, rows_b, A, Bm, gw, gb, stream)
        self.bucket.all_reduce_mean_(self.group)
        return loss

    def _weight_grad(self, N, n, rows_a, rows_b, A, Bm, gw, gb, stream):
        if not self.deterministic:
            L.check(L.lib.qenv_weight_grad(N, n, rows_a, rows_b, L.ptr(A), L.ptr(Bm), L.ptr(gw), L.ptr(gb), stream),
                    "qenv_weight_grad")
            return
        key = (rows_a, rows_b, A.device)
        if key not in self._wg_ws:
            self._wg_ws[key] = torch.empty(int(L.lib.qenv_weight_grad_workspace_bytes(rows_a, rows_b)), dtype=torch.uint8,
                                           device=A.device)
        L.check(L.lib.qenv_weight_grad_deterministic(N, n, rows_a, rows_b, L.ptr(A), L.ptr(Bm), L.ptr(gw), L.ptr(gb),
                                                     L.ptr(self._wg_ws[key]), stream), "qenv_weight_grad_deterministic")

    def _gradients_fused(self, obs, actions, old_logprob, adv, returns, b, valid=None, n_valid=None, n_valid_global=None):
        """The same update with the loss inside the backward kernel: per head one backward launch (which recomputes
        the forward anyway and derives d loss / d output from it) + three weight-gradient launches; no forward launch,
        no loss launch, no [n, r, N] <-> [r, n N] copies."""
        n, K, N = obs.shape
        w, H, stream = self.w, self.w.hidden, _stream(obs.device)
        sums = torch.zeros(5, dtype=torch.float64, device=obs.device)  # [4]: sum_s d actor_s / d std
        std = float(w.std)
        for hid, head in enumerate(("actor", "critic")):
            W = w.head(head)
            n_out = W[4].shape[0]
            if self.one_launch and H == 128 and K == 75:
                # the reference's controller: backward AND the six weight / bias gradients of the head in ONE launch, nothing
                # written to memory in between (qenv_mlp_grad_ppo: the workgroup's four waves share a sample tile and contract it
                # over the samples on the spot).  deterministic: per-workgroup slots added in a fixed order.
                g = [self._state[(head, k)][0] for k in range(6)]
                ws = None
                if self.deterministic:
                    if getattr(self, "_grad_ws", None) is None or self._grad_ws.device != obs.device:
                        self._grad_ws = torch.empty(int(L.lib.qenv_mlp_grad_ppo_workspace_bytes()), dtype=torch.uint8, device=obs.device)
                    ws = self._grad_ws
                L.check(L.lib.qenv_mlp_grad_ppo(N, n, K, H, hid, L.ptr(obs), int(obs.stride(0)), L.ptr(actions), L.ptr(old_logprob), L.ptr(adv),
                                                L.ptr(returns), L.ptr(valid), std, float(self.eps_clip), float(self.c_value),
                                                n * N if valid is None else max(1, int(n_valid_global)), L.ptr(sums),
                                                *[L.ptr(x) for x in W], *[L.ptr(x) for x in g], L.ptr(ws), stream),
                        "qenv_mlp_grad_ppo")
                continue
            L.check(L.lib.qenv_mlp_backward_ppo_masked(N, n, K, H, hid, L.ptr(obs), L.ptr(actions), L.ptr(old_logprob), L.ptr(adv),
                                                       L.ptr(returns), L.ptr(valid), std, float(self.eps_clip), float(self.c_value),
                                                       n * N if valid is None else max(1, int(n_valid_global)),
                                                       L.ptr(sums), *[L.ptr(x) for x in W], L.ptr(b["h1"]), L.ptr(b["h2"]),
                                                       L.ptr(b["dz2"]), L.ptr(b["dz1"]), L.ptr(b["dz3"]), stream),
                    "qenv_mlp_backward_ppo")
            g = [self._state[(head, k)][0] for k in range(6)]
            dz3 = b["dz3"] if n_out == 4 else b["dz3"][:, :n_out].contiguous()
            for A, rows_a, Bm, rows_b, gw, gb in ((dz3, n_out, b["h2"], H, g[4], g[5]), (b["dz2"], H, b["h1"], H, g[2], g[3]),
                                                 (b["dz1"], H, obs, K, g[0], g[1])):
                self._weight_grad(N, n, rows_a, rows_b, A, Bm, gw, gb, stream)
        if valid is None:   # equal shards: every rank's loss is a mean over its own n N samples, the ranks are averaged
            B = float(n * N)
            if self.learn_std:  # before the all-reduce: the std's gradient travels in the same bucket
                self.bucket.flat[-1] = (sums[4] / B - self.c_entropy * 4.0 / std).to(torch.float32)
            self.bucket.all_reduce_mean_(self.group)
        else:               # valid rows: normalised by the GLOBAL count above, so the ranks' contributions add up
            B, world = float(max(1, int(n_valid_global))), self._world()
            if self.learn_std:  # (the entropy term is not a sum over samples: a 1 / world share per rank)
                self.bucket.flat[-1] = (sums[4] / B - self.c_entropy * 4.0 / std / world).to(torch.float32)
            self.bucket.all_reduce_sum_(self.group)
            if world > 1:       # the reported loss terms as the single learner would see them
                import torch.distributed as dist
                dist.all_reduce(sums, group=self.group)
        actor = sums[1] / B
        critic = self.c_value * sums[2] / B
        entropy = torch.full((), -self.c_entropy * 4.0 * (0.5 + 0.9189385332046727 + float(np.log(std))), dtype=torch.float64,
                             device=obs.device)
        return {"loss": actor + critic + entropy, "actor": actor, "critic": critic, "entropy": entropy}

    def adam_step(self):
        self.step += 1
        p = self.flat_p
        L.check(L.lib.qenv_adam_step(p.numel(), L.ptr(p), L.ptr(self.bucket.flat), L.ptr(self.flat_m), L.ptr(self.flat_v),
                                     self.lr, self.betas[0], self.betas[1], self.eps, self.step, _stream(p.device)),
                "qenv_adam_step")
        self.w.touch()
        if self.learn_std:  # the kernels take std by value: one small device-to-host read per Adam step
            self.w.std = float(self.std_param.item())

    def _adam_counted(self):
        p = self.flat_p
        L.check(L.lib.qenv_adam_step_counted(p.numel(), L.ptr(p), L.ptr(self.bucket.flat), L.ptr(self.flat_m), L.ptr(self.flat_v),
                                             self.lr, self.betas[0], self.betas[1], self.eps, L.ptr(self._step_dev),
                                             _stream(p.device)), "qenv_adam_step_counted")

    def _update_graph(self, obs, actions, logprobs, adv, returns):
        """One epoch {zero, backward x2, weight gradients x6, Adam} captured once as a hipGraph and replayed K_epochs times.
        The Adam step count lives in device memory so every replay sees the next one; the captured tensors are the
        rollout's own buffers (a new rollout object / horizon re-captures).  Measured (tools/time_learner_small.py): the
        replay frees the host (no Python / launch work per epoch) but is NOT faster on the device -- 296 vs 281 us per
        epoch at 8 192 samples, 727 vs 702 at 65 536: the epoch is bound by the dependent-kernel latencies of its ~20
        small launches (weight staging, split-K tails), which a graph replays unchanged."""
        # the capture bakes in the buffer pointers AND every by-value hyper-parameter of the epoch's launches
        key = tuple(t.data_ptr() for t in (obs, actions, logprobs, adv, returns)) + tuple(obs.shape) + (
            float(self.w.std), float(self.lr), tuple(self.betas), float(self.eps), float(self.eps_clip), float(self.c_value),
            float(self.c_entropy), bool(self.deterministic))
        dev = obs.device
        if getattr(self, "_graph_key", None) != key:
            if getattr(self, "_step_dev", None) is None:
                self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self.gradients(obs, actions, logprobs, adv, returns)  # eager once: lazily allocated buffers exist afterwards
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=side):
                loss = self.gradients(obs, actions, logprobs, adv, returns)
                self._adam_counted()
                self._graph_loss = torch.stack([loss[k].to(torch.float64).reshape(()) for k in ("loss", "actor", "critic", "entropy")])
            self._graph_key = key
        self._step_dev.fill_(self.step)
        out = []
        for _ in range(self.K_epochs):
            self._graph.replay()
            v = self._graph_loss.clone()
            out.append({"loss": v[0], "actor": v[1], "critic": v[2], "entropy": v[3]})
        self.step += self.K_epochs
        self.w.touch()
        return out

    def update(self, rollout, returns, adv, minibatch=None, generator=None, graph=False, perms=None):
        """K_epochs passes over the collected rollout, returning the list of loss dicts.
        graph=True (full-batch epochs, fixed std, one learner): the epoch is replayed from a hipGraph.
        minibatch=None: full-batch epochs (ppo.py:172-203; the reference's randperm of a full batch does not change
        loss.mean()).  minibatch=B: the schedule of visual_landing/ppo_trainer.py:187-196 -- every epoch walks a fresh
        random permutation in batches of B samples, one Adam step per batch, the remainder as a last smaller batch;
        here the permutation is over the stored time steps and a batch is the B // n_envs steps it covers (all envs
        of a step stay together, so nothing is gathered sample by sample).  A remainder of ONE sample is dropped, as
        `if rest_range > 1` does there (:193).  perms (with minibatch): the K_epochs permutations to walk instead of fresh
        torch.randperm draws -- [K_epochs, n] integers (a recorded schedule: the G26 fixture replays the reference's)."""
        n = rollout.t
        out = []
        valid = getattr(rollout, "valid", None)
        if valid is not None:   # in-band warm-up rollout: the warm-up rows carry weight 0 (full-batch epochs only)
            if minibatch is not None:
                raise ValueError("minibatch epochs over an in-band rollout are not implemented (full-batch epochs are)")
            v32 = valid[:n].to(torch.float32).contiguous()
            n_valid = int(v32.sum().item())
            n_valid_global = self._global_count(n_valid, v32.device)   # once per update: the mask is the same every epoch
            obs = rollout.obs[:n]
            if not obs.is_contiguous() and not (self.fused_loss and self.one_launch and self.w.hidden == 128 and obs.shape[1] == 75):
                obs = obs.contiguous()    # observations stored as slices (PolicyRollout.slice_obs): ONE dense copy for all epochs
                                          # (hidden 128: the one-launch gradient kernel reads the slices in place)
            for _ in range(self.K_epochs):
                out.append(self.gradients(obs, rollout.actions[:n], rollout.logprobs[:n], adv[:n], returns[:n],
                                          valid=v32, n_valid=n_valid, n_valid_global=n_valid_global))
                self.adam_step()
            return out
        if graph and minibatch is None and not self.learn_std and self.fused_loss and not (
                self.group is not False and torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size(self.group if self.group is not None else None) > 1):
            f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else None
            args = [f32(t) for t in (rollout.obs[:n], rollout.actions[:n], rollout.logprobs[:n], adv[:n], returns[:n])]
            if all(a is not None for a in args):  # the captured pointers must be the caller's own (no conversion copies)
                return self._update_graph(*args)
        if minibatch is None:
            for _ in range(self.K_epochs):
                out.append(self.gradients(rollout.obs[:n], rollout.actions[:n], rollout.logprobs[:n], adv[:n], returns[:n]))
                self.adam_step()
            return out
        N = rollout.obs.shape[2]
        per = max(1, int(minibatch) // N)
        if perms is not None:
            perms = torch.as_tensor(perms, device=rollout.obs.device, dtype=torch.int64)
            if tuple(perms.shape) != (self.K_epochs, n):
                raise ValueError(f"perms: expected [{self.K_epochs}, {n}] (one permutation of the stored steps per epoch)")
        for ep in range(self.K_epochs):
            perm = perms[ep] if perms is not None else torch.randperm(n, device=rollout.obs.device, generator=generator)
            rest = n % per
            batches = [perm[k * per:(k + 1) * per] for k in range(n // per)] + ([perm[n - rest:]] if rest * N > 1 else [])
            for idx in batches:
                out.append(self.gradients(rollout.obs[idx], rollout.actions[idx], rollout.logprobs[idx], adv[idx], returns[idx]))
                self.adam_step()
        return out


def visual_reward(total_steps, marker, pos, vel, control, last_shaping, step, ang, v_ang):
    """Batched visual_landing/rl_reward_fuction.py:3-102.  [3, N] tensors (marker, pos, vel,
    control, ang, v_ang), last_shaping [N] (0 = none), step int32 [N].
    Returns (reward [N], shaping [N], done [N] bool, n_solved [N] int32)."""
    N = pos.shape[1]
    dt = L.F64 if pos.dtype == torch.float64 else L.F32
    c = lambda t: t.to(pos.dtype).contiguous()
    marker, pos, vel, control, ang, v_ang, last_shaping = map(c, (marker, pos, vel, control, ang, v_ang, last_shaping))
    step = step.to(torch.int32).contiguous()
    rew = torch.empty(N, dtype=pos.dtype, device=pos.device)
    sh = torch.empty_like(rew)
    done = torch.empty(N, dtype=torch.uint8, device=pos.device)
    ns = torch.empty(N, dtype=torch.int32, device=pos.device)
    stream = _stream(pos.device)
    L.check(L.lib.qenv_visual_reward(dt, N, int(total_steps), L.ptr(marker), L.ptr(pos), L.ptr(vel), L.ptr(control),
                                     L.ptr(last_shaping), L.ptr(step), L.ptr(ang), L.ptr(v_ang), L.ptr(rew), L.ptr(sh),
                                     L.ptr(done), L.ptr(ns), stream), "qenv_visual_reward")
    return rew, sh, done.bool(), ns


class PolicyRollout:
    """Rollout collector of ppo.py's worker loop (:223-269) for N envs at once, all on the device:
    window -> actor + critic (fp32 MFMA) -> Normal sample -> env.step (+ window push) -> masked
    auto-reset of finished episodes, stored time-major; `finish()` runs GAE.

    Every per-step buffer is preallocated and every call enqueues kernels only.  hipGraph capture is valid for the
    DETERMINISTIC policy without auto-reset only (`step(deterministic=True)`, `auto_reset=False`; a replay must cover a
    multiple of T steps so that the ring head returns): the Philox step counter, the ring head and the reset list's
    ping-pong counter are host-side values baked into the kernel arguments, so a replayed sampling / auto-reset step
    would repeat the same noise and reuse a stale list (`capturable` tells).  `env_offset` = this shard's first global env index (keeps the Philox
    streams shard-invariant)."""

    def __init__(self, env: BatchedQuad, weights: ActorCriticWeights, horizon, store_obs=False, auto_reset=False,
                 seed=0, env_offset=0, inband=False, key_by_episode=False, slice_obs=None):
        """inband (with auto_reset): IN-BAND warm-up (qenv_ppo_step, include/qenv.h) -- an env whose episode ends is reset in
        the step kernel's epilogue and the T warm-up steps of that reset are its next T regular steps (zero_control forced,
        window refilled in place); those rows of the rollout get `valid` = 0 and are left out of GAE's moments and of the
        loss (finish() and PpoLearner.update take care of it).  No warm-up launch: a step is policy + env step.  The valid
        rows of an env are the samples the default schedule collects for the same episodes, T rows later per restart.
        key_by_episode: an env's n-th restart draws its state from the stream of n, not of the step it happens in.
        slice_obs (None = whenever possible: in-band schedule, stored observations, hidden 128, T = 5): the observations are
        stored as ONE 15-float slice per step -- under the in-band schedule every env pushes exactly one slice per step, so
        the windows of consecutive steps overlap in T - 1 slices; `obs` is then a strided VIEW [horizon, 15 T, N] of the
        slice array [(horizon + T - 1) * 15, N] (same indexing, a fifth of the memory and of the store traffic; the
        critic reads it in place, the learner takes one contiguous copy per update)."""
        assert env.direct_control_flag, "the PPO controller commands the four rotors (direct_control=1)"
        self.env, self.w, self.horizon = env, weights, int(horizon)
        self.auto_reset, self.seed, self.env_offset = bool(auto_reset), int(seed), int(env_offset)
        self.inband, self.key_by_episode = bool(inband), bool(key_by_episode)
        if self.inband and not self.auto_reset:
            raise ValueError("inband=True is a schedule of the auto-reset rollout (auto_reset=True)")
        self.keep_init = False  # True: auto-resets also leave the drawn initial states in `self.init`
        # With the observations stored, V(s_t) of every step is evaluated in ONE launch over the whole horizon at
        # finish() instead of one launch per step: same kernel, same arithmetic (bit-identical values), but each
        # workgroup stages the 119 KB of weights once per horizon, not once per step (45 -> 31 us per 65 536 envs)
        self.batched_critic = bool(store_obs and weights.critic is not None and env.T == 5
                                   and weights.hidden in (16, 32, 64, 128))
        self._values_f32 = None
        self.fused_act = True  # actor forward and Gaussian sampling as one launch (qenv_policy_act); False: two launches
        # auto-reset steps as ONE library call (qenv_ppo_step: the reset happens in the step kernel's epilogue, the warm-up
        # of the envs that finished stays pending until the next step / finish() / env.ppo_flush(); with
        # QENV_PPO_OVERLAP=1 that warm-up runs on a side stream beside the next policy step).  Bit-identical buffers.
        self.overlap = True
        N, dev, dt, H = env.n_envs, env.device, env.dtype, self.horizon
        self.actions = torch.zeros(H, 4, N, device=dev, dtype=dt)
        self.logprobs = torch.zeros(H, 4, N, device=dev, dtype=dt)
        self.rewards = torch.zeros(H, N, device=dev, dtype=dt)
        self.masks = torch.ones(H, N, device=dev, dtype=dt)
        self.valid = torch.ones(H, N, device=dev, dtype=dt) if self.inband else None   # 0: a restarted env's warm-up row
        self.values = torch.zeros(H + 1, N, device=dev, dtype=dt)
        can_slice = bool(store_obs and self.inband and env.T == 5 and weights.hidden == 128)
        if slice_obs and not can_slice:
            raise ValueError("slice_obs needs store_obs, the in-band schedule, hidden 128 and T = 5")
        self.slice_obs = can_slice if slice_obs is None else bool(slice_obs)
        if self.slice_obs:
            self._obs_slices = torch.zeros((H + env.T - 1) * 15, N, device=dev, dtype=torch.float32)
            self.obs = torch.as_strided(self._obs_slices, (H, 15 * env.T, N), (15 * N, N, 1))
        else:
            self.obs = torch.zeros(H, 15 * env.T, N, device=dev, dtype=torch.float32) if store_obs else None
        self.mean = torch.zeros(4, N, device=dev, dtype=dt)
        self.store_mean = False   # True: the fused rollout step also writes the policy mean of its last step here (4 more row stores)
        self.done = torch.zeros(N, device=dev, dtype=torch.uint8)
        self.init = torch.zeros(13, N, device=dev, dtype=dt)
        self.t = 0
        self.global_step = 0

    @property
    def capturable(self):
        """True if step(deterministic=True) may be captured into a hipGraph (see the class docstring)."""
        return not self.auto_reset

    def start(self, init_states=None):
        """reset() + fill the window with the T warm-up slices (ppo.py:234-240)."""
        env = self.env
        if init_states is None:
            init_states = env.sample_init_states(self.seed, 0, self.env_offset, out=self.init)
        if env.robust_control:  # domain randomisation: a new plant per episode, Philox gusts (quadrotor_env.py:84-109)
            env.robust_gusts(None, self.seed, self.env_offset)
            env.robust_params(None, self.seed, self.env_offset)
        env.reset(init_states, return_history=False, push_window=True)
        self.t = 0

    def step(self, noise=None, deterministic=False):
        """One env step for all envs.  noise: injected [4, N] standard normal."""
        env, t, w = self.env, self.t, self.w
        store = None if self.obs is None else self.obs[t]
        fused = (self.auto_reset and self.overlap and self.fused_act and not deterministic and not self.keep_init
                 and not env.robust_control and (w.critic is None or self.batched_critic))
        if self.inband and not fused:
            raise ValueError("inband=True needs the fused step (stochastic policy, stored observations or no critic, no "
                             "robust_control, keep_init off)")
        if fused:
            env.ppo_step(w, self.seed, self.global_step + 1, self.env_offset, self.actions[t], self.logprobs[t],
                         self.mean if self.store_mean else None,
                         store, self.rewards[t], self.done, self.masks[t], noise=noise,
                         valid=None if self.valid is None else self.valid[t], inband=self.inband,
                         key_by_episode=self.key_by_episode,
                         obs_rows_from=15 * (env.T - 1) if (self.slice_obs and t > 0) else 0)
            self.t += 1
            self.global_step += 1
            return
        if deterministic:  # mean action, as ppo_quad_eval.py:55
            act = env.mlp_forward(w, "actor", out=self.actions[t], obs_store=store)
        elif self.fused_act:  # forward + sample in one launch
            act = env.policy_act(w, noise, self.seed, self.global_step + 1, self.env_offset, out=self.actions[t],
                                 logprob=self.logprobs[t], mean=self.mean, obs_store=store)
        else:
            env.mlp_forward(w, "actor", out=self.mean, obs_store=store)
            act = env.sample_action(self.mean, w.std, noise, self.seed, self.global_step + 1, self.env_offset,
                                    out=self.actions[t], logprob=self.logprobs[t])
        if w.critic is not None and not self.batched_critic:
            env.mlp_forward(w, "critic", out=self.values[t:t + 1])
        env.step(act, flags=L.STEP_NO_AUX, out=(None, self.rewards[t], self.done), out_mask=self.masks[t],
                 push_window=True)
        if self.auto_reset:  # worker loop: `while` episodes until the batch is full (ppo.py:232-262)
            if env.robust_control:
                env.robust_params(None, self.seed, self.env_offset, mask=self.done)
            if self.keep_init:  # tests compare against a fresh twin reset to the same states
                env.sample_init_states(self.seed, self.global_step + 1, self.env_offset, mask=self.done, out=self.init)
                env.reset(self.init, mask=self.done, return_history=False, push_window=True)
            else:  # one launch: states drawn inside the reset kernel from the same Philox streams
                env.reset_sampled(self.seed, self.global_step + 1, self.env_offset, mask=self.done, push_window=True)
        self.t += 1
        self.global_step += 1

    def run(self, n_steps, noise=None):
        """n_steps rollout steps from slot self.t on in ONE library call (qenv_ppo_rollout: the worker loop's `for t` in C):
        the same buffers as n_steps step() calls, bit for bit, without a Python round trip between the launches.  The
        sampling auto-reset rollout only (what step() runs as the fused call); noise: injected [n_steps, 4, N]."""
        env, w, t0 = self.env, self.w, self.t
        n_steps = int(n_steps)
        if n_steps <= 0:
            return
        if t0 + n_steps > self.horizon:
            raise ValueError(f"run({n_steps}) from slot {t0} exceeds the horizon {self.horizon}")
        fused = (self.auto_reset and self.overlap and self.fused_act and not self.keep_init and not env.robust_control
                 and (w.critic is None or self.batched_critic))
        if not fused:
            for k in range(n_steps):
                self.step(noise=None if noise is None else noise[k])
            return
        nz = None
        if noise is not None:
            nz = noise.to(device=env.device, dtype=env.dtype).contiguous()
            if tuple(nz.shape) != (n_steps, 4, env.n_envs):
                raise ValueError(f"noise: expected [{n_steps}, 4, {env.n_envs}]")
        key = (id(w), w.actor[0].data_ptr())
        if getattr(env, "_ppo_checked", None) != key:
            env._check_head(w.head("actor"), "actor", n_out=4)
            env._ppo_checked = key
        env._ensure_prepared(w, "actor")
        W = w.head("actor")
        a = L.PpoStepArgs(int(W[0].shape[1]), int(W[0].shape[0]), *[x.data_ptr() for x in W], float(w.std),
                          None if nz is None else nz.data_ptr(), int(self.seed), int(self.global_step + 1), int(self.env_offset),
                          self.mean.data_ptr() if self.store_mean else None, self.actions[t0].data_ptr(), self.logprobs[t0].data_ptr(),
                          None if self.obs is None else self.obs[t0].data_ptr(), self.rewards[t0].data_ptr(), self.done.data_ptr(),
                          self.masks[t0].data_ptr(), None if self.valid is None else self.valid[t0].data_ptr(), int(self.inband),
                          int(self.key_by_episode), 15 * (env.T - 1) if (self.slice_obs and t0 > 0) else 0, int(self.slice_obs))
        env._bind()
        L.check(L.lib.qenv_ppo_rollout(env._h, C.byref(a), n_steps), "qenv_ppo_rollout")
        self.t += n_steps
        self.global_step += n_steps

    def run_deterministic(self, n_steps):
        """n_steps steps of the DETERMINISTIC policy (mean action, ppo_quad_eval.py:55) from slot self.t on in ONE library call
        (qenv_policy_rollout): the buffers of n_steps step(deterministic=True) calls, bit for bit.  No auto-reset, no stored
        observations, no critic (the evaluation loop)."""
        n_steps = int(n_steps)
        if n_steps <= 0:
            return
        t0 = self.t
        if t0 + n_steps > self.horizon:
            raise ValueError(f"run_deterministic({n_steps}) from slot {t0} exceeds the horizon {self.horizon}")
        if self.auto_reset or self.obs is not None or (self.w.critic is not None and not self.batched_critic):
            for _ in range(n_steps):
                self.step(deterministic=True)
            return
        self.env.policy_rollout(self.w, n_steps, self.actions[t0:t0 + n_steps], self.rewards[t0:t0 + n_steps], self.done,
                                self.masks[t0:t0 + n_steps])
        self.t += n_steps
        self.global_step += n_steps

    def critic_values(self, n=None):
        """values[:n] = critic(obs[:n]) in one launch over the stored observations (batched_critic)."""
        n = self.t if n is None else n
        env, W = self.env, self.w.critic
        out = self.values[:n]
        if out.dtype != torch.float32:  # fp64 handles: the kernel's float result, widened (as the per-step launch does)
            if self._values_f32 is None:
                self._values_f32 = torch.empty(self.horizon, env.n_envs, device=env.device, dtype=torch.float32)
            out = self._values_f32[:n]
        L.check(L.lib.qenv_dense_tail_chunks_strided(env.n_envs, n, 15 * env.n_envs if self.slice_obs else 0, 15 * env.T,
                                                     self.w.hidden, 1, 0, L.ptr(self.obs), *[L.ptr(x) for x in W], L.ptr(out),
                                                     _stream(env.device)), "qenv_dense_tail_chunks_strided")
        if out.dtype != self.values.dtype:
            self.values[:n].copy_(out)
        return self.values[:n]

    def finish(self, ddof=0, group=False, bootstrap=False, deterministic=False, normalize=True, return_stats=False):
        """GAE over the collected steps.  bootstrap=False appends V=0 like the reference
        (ppo.py:385); True evaluates the critic on the final window.  normalize / return_stats: as `gae`."""
        n = self.t
        self.env.ppo_flush()  # the envs that finished in the last step: their warm-up before anything reads the env
        if self.batched_critic and n:
            self.critic_values(n)
        if bootstrap and self.w.critic is not None:
            self.env.mlp_forward(self.w, "critic", out=self.values[n:n + 1])
        else:
            self.values[n].zero_()
        return gae(self.values[:n + 1], self.masks[:n], self.rewards[:n], ddof=ddof, group=group, deterministic=deterministic,
                   normalize=normalize, return_stats=return_stats, valid=None if self.valid is None else self.valid[:n])


class StreamShardedRollout:
    """One GPU's env range collected as S independent shards on S HIP streams (each shard its own env handle and
    `PolicyRollout`).  The rollout step of a shard ends in the latency-bound warm-up of its finished envs -- a handful of
    wavefronts for ~60 us -- during which a single batch leaves the chip idle; with two shards of half the size the other
    shard's policy forward and env step run meanwhile (their workgroups cover half the CUs each): 137 -> 118 us per step
    at 65 536 envs (tools/time_rollout_shards.py; four shards lose again to launch overhead).  Env trajectories, Philox
    streams and rollout buffers are shard-invariant (keyed by the global env index), so the union of the shards is the
    single-batch run bit for bit; GAE's normalisation moments are summed over the shards (and all-reduced over `group`)
    before any shard normalises, so the advantages are those of one learner over the whole range.

    make_env(lo, hi) -> BatchedQuad for the local env range [lo, hi) (already reset if the caller wants its own
    initial states; otherwise call start())."""

    def __init__(self, make_env, n_envs, weights, horizon, shards=2, seed=0, env_offset=0, device=None, **kw):
        self.n_envs, self.shards = int(n_envs), int(shards)
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.shards)]
        self.bounds = [(k * self.n_envs // self.shards, (k + 1) * self.n_envs // self.shards) for k in range(self.shards)]
        self.ros = []
        main = torch.cuda.current_stream(dev)
        for st, (lo, hi) in zip(self.streams, self.bounds):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self.ros.append(PolicyRollout(make_env(lo, hi), weights, horizon, seed=seed, env_offset=env_offset + lo, **kw))
        if self.shards > 1:
            # the shards' fused rollout launches are meant to run side by side: 64-env blocks keep a shard to its share of the
            # chip (with the 32-env blocks a small shard would pick by itself every launch fills the chip and they serialise)
            # (the block size only: a handle switched to the separate launches -- QENV_PPO_FUSED=0, set_ppo_fused(False) -- stays so)
            for ro in self.ros:
                ro.env.set_ppo_fused_block(64)
        self.w = weights
        self._stats = torch.zeros(3, dtype=torch.float64, device=dev)
        self._fast = None   # built at the first sampling step (see _fast_args)

    @property
    def t(self):
        return self.ros[0].t

    @t.setter
    def t(self, v):
        for ro in self.ros:
            ro.t = v

    def start(self):
        for st, ro in zip(self.streams, self.ros):
            with torch.cuda.stream(st):
                ro.start()

    def _fast_args(self):
        """One prebuilt qenv_ppo_step argument block per (shard, time index): the sampling auto-reset rollout is one C
        call per shard and step, and with several shards the Python around it (stream context, buffer checks, struct
        assembly: ~60 us per shard-step) becomes the limit -- here a step costs one ctypes call per shard."""
        import ctypes as C
        out = []
        for st, ro in zip(self.streams, self.ros):
            env, w = ro.env, ro.w
            if not (ro.auto_reset and ro.overlap and ro.fused_act and not ro.keep_init and not env.robust_control
                    and (w.critic is None or ro.batched_critic)):
                return None
            with torch.cuda.stream(st):
                env._check_head(w.head("actor"), "actor", n_out=4)
                env._bind()  # the handle keeps this shard's stream from here on
            W = w.head("actor")
            per_t = []
            for t in range(ro.horizon):
                per_t.append(L.PpoStepArgs(int(W[0].shape[1]), int(W[0].shape[0]), *[x.data_ptr() for x in W], float(w.std), None,
                                           int(ro.seed), 0, int(ro.env_offset), ro.mean.data_ptr() if ro.store_mean else None,
                                           ro.actions[t].data_ptr(),
                                           ro.logprobs[t].data_ptr(), None if ro.obs is None else ro.obs[t].data_ptr(),
                                           ro.rewards[t].data_ptr(), ro.done.data_ptr(), ro.masks[t].data_ptr(),
                                           None if ro.valid is None else ro.valid[t].data_ptr(), int(ro.inband),
                                           int(ro.key_by_episode), 15 * (env.T - 1) if (ro.slice_obs and t > 0) else 0, 0))
            out.append((env._h, per_t, float(w.std)))
        self._C = C
        return out

    def _join_consumer(self):
        """First step of a horizon: the shard streams wait for everything the caller's stream has queued on the buffers
        the next horizon overwrites (a learner update, an all-gather of the returns, adv_normalize of the last finish())."""
        main = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(main)

    def step(self, deterministic=False):
        """One step of every shard.  The rollout buffers (obs / actions / logprobs / rewards / masks / values) and the
        tensors finish() returned are overwritten from here on: consume them on the caller's stream BEFORE the first step
        of the next horizon (that step makes the shard streams wait for the caller's stream)."""
        if self.ros[0].t == 0:
            self._join_consumer()
        if not deterministic:
            if self._fast is None:
                self._fast = self._fast_args() or False
            if self._fast:
                C = self._C
                for (h, per_t, std), ro, st in zip(self._fast, self.ros, self.streams):
                    if not ro.env._prepared_is_current(ro.w, "actor"):   # rebuilt on the SHARD's stream (the handle stays bound to it)
                        with torch.cuda.stream(st):
                            ro.env._ensure_prepared(ro.w, "actor")
                    a = per_t[ro.t]
                    a.step = ro.global_step + 1
                    a.std = float(ro.w.std)  # always: a learner / schedule may move the std (and move it back) between steps
                    L.check(L.lib.qenv_ppo_step(h, C.byref(a)), "qenv_ppo_step")
                    ro.t += 1
                    ro.global_step += 1
                return
        for st, ro in zip(self.streams, self.ros):
            with torch.cuda.stream(st):
                ro.step(deterministic=deterministic)

    def run(self, n_steps):
        """n_steps steps of every shard, one library call per shard (PolicyRollout.run) on the shard's stream: the host
        enqueues a shard's whole horizon at once and the streams interleave on the device."""
        if self.ros[0].t == 0:
            self._join_consumer()
        for st, ro in zip(self.streams, self.ros):
            with torch.cuda.stream(st):
                ro.run(n_steps)

    def finish(self, ddof=0, group=None, bootstrap=False, deterministic=False):
        """Per shard: pending warm-ups, critic over the horizon, GAE; then ONE normalisation over all shards (and ranks).
        Returns [(returns, advantages)] per shard, valid on the caller's current stream."""
        main = torch.cuda.current_stream(self.device)
        outs = []
        for st, ro in zip(self.streams, self.ros):
            with torch.cuda.stream(st):
                outs.append(ro.finish(ddof=ddof, group=False, bootstrap=bootstrap, deterministic=deterministic,
                                      normalize=False, return_stats=True))
            main.wait_stream(st)
            for t in outs[-1]:  # allocated on the shard's stream, consumed on the caller's: the caching allocator must not
                t.record_stream(main)  # hand the block to the shard's next gae() while the caller's stream still reads it
        torch.sum(torch.stack([o[2] for o in outs]), dim=0, out=self._stats)
        if group is not False and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size(group) > 1:
            torch.distributed.all_reduce(self._stats, group=group)
        for o, ro in zip(outs, self.ros):
            adv_normalize(o[1], self._stats, ddof, valid=None if ro.valid is None else ro.valid[:ro.t])
        return [(o[0], o[1]) for o in outs]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def last_returns(self):
        """The banked return of every env's last finished episode, local env order (for the all-gather)."""
        return torch.cat([ro.env.last_return for ro in self.ros])
