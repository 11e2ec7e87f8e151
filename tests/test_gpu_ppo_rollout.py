"""GPU: the PPO rollout path -- window push fused into the step, action / initial-state samplers,
the rollout collector against the CPU oracle with injected noise, masked auto-reset, GAE."""
import numpy as np
import pytest
import torch

from .conftest import golden, rel_err, scaled_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aq():
    import autonomous_quadrotor_environment_amd as aq
    assert torch.cuda.is_available()
    return aq


def _np(t):
    return t.detach().double().cpu().numpy()


def _weights(aq, N=128):
    g = golden(f"g9_mlp_N{N}")
    sd = {k: g[k] for k in g.files if k.startswith(("actor", "critic"))}
    return aq.ActorCriticWeights.from_state_dict(sd), g


def _ordered_window(env):
    """ring -> [N, 15*T] in time order (oldest first), like dl_in_gen.deep_learning_input."""
    T = env.T
    ring = _np(env.window).reshape(T, 15, env.n_envs)
    head = env._pushes % T if hasattr(env, "_pushes") else 0
    return ring


def test_step_push_window_equals_separate_push(aq):
    n = 512
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    x0 = torch.from_numpy(fast_random_init_states(1, n).T.copy())
    a = aq.BatchedQuad(n, 0.01, 1000, training=False, direct_control=1, T=5)
    b = aq.BatchedQuad(n, 0.01, 1000, training=False, direct_control=1, T=5)
    sa, aa = a.reset(x0)
    for k in range(5):
        a.window_push(sa[k], aa[k], reset_first=(k == 0))
    b.reset(x0, push_window=True)
    assert torch.equal(a.window, b.window)
    rng = np.random.RandomState(0)
    for k in range(7):
        act = torch.from_numpy(rng.normal(0, 0.6, (4, n))).float().cuda()  # unclipped on purpose
        obs, _, _ = a.step(act)
        a.window_push(obs, act)
        b.step(act, push_window=True)
        assert torch.equal(a.window, b.window), k
        assert torch.equal(a.state, b.state)


def test_sample_action_injected_and_philox(aq):
    n = 100000
    env = aq.BatchedQuad(n, 0.01, 1000, direct_control=1, T=1, dtype=torch.float32)  # fp32 buffers below
    mean = torch.randn(4, n, device="cuda")
    noise = torch.randn(4, n, device="cuda")
    lp = torch.empty(4, n, device="cuda")
    act = env.sample_action(mean, 0.1, noise=noise, logprob=lp)
    assert torch.allclose(act, mean + 0.1 * noise, rtol=0, atol=1e-6)
    ref_lp = torch.distributions.Normal(mean, 0.1).log_prob(act)
    assert torch.allclose(lp, ref_lp, rtol=0, atol=2e-3) and float((lp - ref_lp).abs().mean()) < 2e-5
    # Philox path: N(0,1) moments, determinism, step/seed sensitivity, shard invariance
    z = (env.sample_action(torch.zeros_like(mean), 1.0, seed=7, step=3) ).double()
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1) < 0.01
    assert abs(float((z ** 4).mean()) - 3) < 0.1
    assert abs(float(torch.corrcoef(z)[0, 1])) < 0.01
    assert torch.equal(z.float(), env.sample_action(torch.zeros_like(mean), 1.0, seed=7, step=3))
    assert not torch.equal(z.float(), env.sample_action(torch.zeros_like(mean), 1.0, seed=7, step=4))
    assert not torch.equal(z.float(), env.sample_action(torch.zeros_like(mean), 1.0, seed=8, step=3))
    half = aq.BatchedQuad(n // 2, 0.01, 1000, direct_control=1, T=1, dtype=torch.float32)
    zh = half.sample_action(torch.zeros(4, n // 2, device="cuda"), 1.0, seed=7, step=3, env_offset=n // 2)
    assert torch.equal(zh, z.float()[:, n // 2:])


def test_sample_init_states_distribution(aq):
    n = 200000
    env = aq.BatchedQuad(n, 0.01, 1000, direct_control=1, T=1, dtype=torch.float32)  # fp32 buffers below
    x = _np(env.sample_init_states(11, episode=2))
    assert np.allclose(np.linalg.norm(x[6:10], axis=0), 1, atol=1e-6)
    assert np.all(np.abs(x[0:5:2]) <= 2.5) and np.all(np.abs(x[1:6:2]) <= 5)
    assert np.all(x[10:13] <= 7.5) and np.all(x[10:13] >= -15)
    # clipped N(0,2): P(|z|>1.25 sigma) = 0.2113 sits at the position clip
    assert abs((np.abs(x[0]) >= 2.5 - 1e-6).mean() - 0.2113) < 0.01
    assert abs(x[1].std() - np.sqrt(np.mean(np.clip(np.random.RandomState(0).normal(0, 2, 10 ** 6), -5, 5) ** 2))) < 0.02
    # Euler angles U(-.5,.5) through euler_quat
    from autonomous_quadrotor_environment_amd import quat_euler
    ang = np.array([quat_euler(x[6:10, k]) for k in range(2000)])
    assert np.all(np.abs(ang) <= 0.5 + 1e-5) and abs(ang.std() - 1 / np.sqrt(12)) < 0.02
    # mask + shard invariance + episode sensitivity
    m = torch.zeros(n, dtype=torch.uint8, device="cuda"); m[::3] = 1
    out = torch.full((13, n), 9.0, device="cuda")
    env.sample_init_states(11, episode=2, mask=m, out=out)
    o = _np(out)
    assert np.array_equal(o[:, ::3], x[:, ::3].astype(np.float32)) and np.all(o[:, 1::3] == 9)
    half = aq.BatchedQuad(n // 2, 0.01, 1000, direct_control=1, T=1, dtype=torch.float32)
    assert np.array_equal(_np(half.sample_init_states(11, 2, env_offset=n // 2)), x[:, n // 2:])
    assert not np.array_equal(_np(env.sample_init_states(11, episode=3)), x)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_policy_rollout_vs_oracle_with_injected_noise(aq, oracle, dt):
    """16 envs x 40 steps of the stochastic policy (actor + critic N=128, std 0.1) with the SAME
    noise tensor fed to the HIP path and to the fp64 oracle (env + window + MLP + GAE)."""
    w, g = _weights(aq, 128)
    w.std = 0.1
    n, steps = 16, 40
    x0 = oracle.random_init_states(31, n)
    rng = np.random.RandomState(32)
    noise = rng.normal(0, 1, (steps, 4, n))
    env = aq.BatchedQuad(n, 0.01, 1000, training=False, direct_control=1, T=5, dtype=dt)
    ro = aq.PolicyRollout(env, w, steps, store_obs=True)
    ro.start(torch.from_numpy(x0.T.copy()))
    for k in range(steps):
        ro.step(noise=torch.from_numpy(noise[k]))
    ret, adv = ro.finish(ddof=0, group=False)
    A = [g[f"actor_{l}_{p}"] for l in (0, 2, 4) for p in ("weight", "bias")]
    Cw = [g[f"critic_{l}_{p}"] for l in (0, 2, 4) for p in ("weight", "bias")]
    tol = 2e-3 if dt == torch.float32 else 2e-4  # the MLP runs in fp32 in both handles
    R, V, M = np.zeros((steps, n)), np.zeros((steps + 1, n)), np.ones((steps, n))
    for e in range(n):
        ref = oracle.OracleQuad(0.01, 1000, training=False, direct_control=1, T=5)
        s, a = ref.reset(x0[e])
        win = oracle.OracleWindow(5)
        obs = win.dl_input(s, a)
        for k in range(steps):
            assert np.max(np.abs(_np(ro.obs[k, :, e]) - obs)) < tol, (e, k)
            mean = oracle.mlp_forward(obs.astype(np.float64), *A, True)[0]
            V[k, e] = oracle.mlp_forward(obs.astype(np.float64), *Cw, False)[0, 0]
            act = mean + 0.1 * noise[k, :, e]
            assert np.max(np.abs(_np(ro.actions[k, :, e]) - act)) < tol, (e, k)
            qs, r, d = ref.step(act)
            R[k, e] = r; M[k, e] = 0.0 if d else 1.0
            obs = win.dl_input(qs, act[None, :])
        assert scaled_err(_np(env.state[:, e]), ref.state) < tol
    assert np.max(np.abs(_np(ro.rewards) - R)) < 5e-2 * tol / 2e-3 + 1e-3
    assert np.array_equal(_np(ro.masks), M)
    assert rel_err(_np(ro.values[:steps]), V[:steps], 1.0) < tol
    ret_ref = np.zeros((steps, n)); adv_ref = np.zeros((steps, n))
    for e in range(n):
        ret_ref[:, e], adv_ref[:, e] = oracle.gae(np.append(V[:steps, e], 0.0), M[:, e], R[:, e], ddof=0)
    assert rel_err(_np(ret), ret_ref, 1.0) < 5 * tol
    # the normalisation is global over all envs and steps
    raw = ret_ref - V[:steps]
    adv_glob = (raw - raw.mean()) / (raw.std() + 1e-10)
    assert rel_err(_np(adv), adv_glob, 1.0) < 5 * tol


def test_auto_reset_keeps_rings_in_phase_and_banks_returns(aq):
    """training=True with a short horizon: envs time out at different... the same step here, and
    some leave the bounding box earlier; every reset env must look exactly like a fresh env reset
    to the same initial state, while the others are untouched."""
    w, _ = _weights(aq, 128)
    n, T = 256, 5
    env = aq.BatchedQuad(n, 0.01, 12, training=True, direct_control=1, T=T)
    ro = aq.PolicyRollout(env, w, 40, auto_reset=True, seed=5)
    ro.start()
    ep_before = env.episodes.clone()
    returns_running = torch.zeros(n, device="cuda")
    for k in range(40):
        prev_ret = env.episode_return.clone()
        ro.step(deterministic=True)
        done = ro.done.bool()
        # banked return = running return of the finished episode (including this step's reward)
        if bool(done.any()):
            banked = env.last_return[done]
            assert torch.allclose(banked, (prev_ret + ro.rewards[k])[done], atol=1e-4)
            # fresh twin: same init states, reset with window push
            # (the auto-reset draws the states inside the reset kernel; the separate sampler has the same streams)
            twin = aq.BatchedQuad(n, 0.01, 12, training=True, direct_control=1, T=T)
            twin.reset(env.sample_init_states(5, k + 1), push_window=True)
            assert torch.equal(env.state[:, done], twin.state[:, done])
            assert torch.equal(env.i[done], twin.i[done]) and int(env.i[done].max()) == T
            # ring: env's slots are rotated by its head relative to the twin's (head 0)
            head = (T + k + 1) % T  # T pushes at start + (k+1) step pushes
            ring_env = env.window.view(T, 15, n)[:, :, done]
            ring_twin = twin.window.view(T, 15, n)[:, :, done]
            order = [(head + j) % T for j in range(T)]
            assert torch.equal(ring_env[order], ring_twin)
    assert int((env.episodes - ep_before).min()) >= 2  # n_max=12 (+T): every env timed out at least twice
    assert bool(torch.isfinite(env.state).all())
    assert float(ro.masks.min()) == 0.0 and float(ro.masks.max()) == 1.0


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_ppo_loss_and_seed_gradients_golden(aq, dt):
    """G15: loss of PPO.update (ppo.py:185-203) and d loss / d mean, d loss / d value against the reference's
    evaluate() + torch autograd."""
    g = golden("g15_ppo_loss")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    out = aq.ppo_loss(dev(g["mean"].T), dev(g["value"]), dev(g["actions"].T), dev(g["old_logprob"].T), dev(g["adv"]),
                      dev(g["returns"]), float(g["std"]), with_grads=True)
    tol = 2e-6 if dt == torch.float64 else 2e-5  # the reference itself mixes an fp32 scale tensor in (model.py:79)
    for k in ("loss", "actor", "critic", "entropy"):
        assert abs(float(out[k]) - float(g[k])) < tol * max(1.0, abs(float(g[k]))), k
    assert np.max(np.abs(_np(out["grad_mean"]).T - g["grad_mean"])) < tol * 10
    assert np.max(np.abs(_np(out["grad_value"]) - g["grad_value"])) < tol
    assert float((out["grad_mean"] == 0).float().mean()) > 0.02  # some samples sit on the saturated clamp


def w_small():
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *sh: torch.randn(*sh, device="cuda", generator=g) * 0.05
    return [r(64, 800), r(64), r(64, 64), r(64), r(3, 64), r(3)]


def test_dense_tail_rl_nn_model_shape(aq, oracle):
    """rl_nn_model.py dense tail 800 -> 64 -> 64 -> 3 (tanh x3) and its critic twin (-> 1, no final tanh) on
    synthetic U(-1,1) features, against the fp64 oracle MLP; ragged N."""
    rng = np.random.RandomState(9)
    # (800, 64, ., ., 65 541): past the 2 048 sample tiles from which H = 64 runs k_dense_tail_v3 (split-bf16 matrix cores, W1 streamed
    # as ready fragments), ragged last tile; (72, 64, ...): a K that is not a multiple of the 64-k chunk
    for K, H, n_out, ft, N in ((800, 64, 3, True, 1000), (800, 64, 1, False, 37), (130, 32, 4, True, 4096), (75, 128, 4, True, 333),
                               (800, 64, 3, True, 65541), (72, 64, 1, False, 65600)):
        def lin(o, i):
            b = 1.0 / np.sqrt(i)
            return rng.uniform(-b, b, (o, i)), rng.uniform(-b, b, o)
        W1, b1 = lin(H, K); W2, b2 = lin(H, H); W3, b3 = lin(n_out, H)
        x = rng.uniform(-1, 1, (K, N)).astype(np.float32)
        w = [torch.from_numpy(v).float().cuda() for v in (W1, b1, W2, b2, W3, b3)]
        got = _np(aq.dense_tail(torch.from_numpy(x).cuda(), w, final_tanh=ft)).T
        ref = oracle.mlp_forward(x.T.astype(np.float64), W1.astype(np.float32).astype(np.float64), b1.astype(np.float32).astype(np.float64),
                                 W2.astype(np.float32).astype(np.float64), b2.astype(np.float32).astype(np.float64),
                                 W3.astype(np.float32).astype(np.float64), b3.astype(np.float32).astype(np.float64), ft)
        assert got.shape == (N, n_out)
        assert np.max(np.abs(got - ref)) < 5e-6, (K, H, n_out, np.max(np.abs(got - ref)))
        if H == 64 and K % 8 == 0 and N >= 65536:
            # the features in the reference's own layout ([batch, K], rl_nn_model.py:58): the same bits, no transposed copy
            rows = _np(aq.dense_tail(torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), w, final_tanh=ft, rows=True)).T
            assert np.array_equal(rows, got), (K, N)
    with pytest.raises(aq._lib.QenvError):   # the row layout is read by k_dense_tail_v3 only: small batches say so
        aq.dense_tail(torch.zeros(1000, 800, device="cuda"), w_small(), final_tanh=True, rows=True)


def test_dense_tail_landing_model_fixture(aq):
    """a24 pinned by its own class: k_dense_tail on the 800 features that visual_landing/rl_nn_model.py's ActorCritic
    conv front-end produced, against what the class's own actor / critic forward returned (g22; fp64 twin of the same
    weights and the class's fp32 result)."""
    g = golden("g22_landing_dense_tail")
    for head, ft in (("actor", True), ("critic", False)):
        w = [torch.from_numpy(g[f"{head}_{n}{k}"]).float().cuda() for k in (1, 2, 3) for n in ("W", "b")]
        x = torch.from_numpy(np.ascontiguousarray(g[f"feat_{head}"].T)).cuda()   # [800, 48] feature-major
        got = _np(aq.dense_tail(x, w, final_tanh=ft)).T
        assert got.shape == g[f"out_{head}64"].shape
        assert np.max(np.abs(got - g[f"out_{head}64"])) < 2e-6, head
        assert np.max(np.abs(got - g[f"out_{head}32"])) < 3e-6, head


def _torch_nets(w, dtype=torch.float64):
    import torch.nn as nn
    def net(ws, final_tanh):
        W1, b1, W2, b2, W3, b3 = ws
        l = [nn.Linear(W1.shape[1], W1.shape[0]), nn.Tanh(), nn.Linear(W2.shape[1], W2.shape[0]), nn.Tanh(),
             nn.Linear(W3.shape[1], W3.shape[0])] + ([nn.Tanh()] if final_tanh else [])
        m = nn.Sequential(*l).to("cuda", dtype)
        with torch.no_grad():
            for lin, (W, b) in zip([m[0], m[2], m[4]], ((W1, b1), (W2, b2), (W3, b3))):
                lin.weight.copy_(W.to(dtype)); lin.bias.copy_(b.to(dtype))
        return m
    return net(w.actor, True), net(w.critic, False)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("hidden,shape", [(16, (3, 200)), (32, (3, 200)), (64, (3, 200)), (128, (3, 200)), (128, (8, 5000))])
def test_ppo_update_gradients_and_adam_vs_torch_autograd(aq, hidden, shape, fused):
    """Backward pass (MFMA), weight/bias gradients (split-K GEMM) and the Adam step of PpoLearner against torch
    autograd + torch.optim.Adam on the reference's loss (ppo.py:185-203), fp64 torch as the yardstick.  fused: the
    loss evaluated inside the backward kernel (qenv_mlp_backward_ppo, the default) or forward / qenv_ppo_loss /
    backward as separate launches.  The (8, 5000) shape is past the 1 024 sample tiles from which the N = 128 network runs
    k_mlp_bwd_v3 (split-bf16 matrix cores, one W2 image for the row reads and the transposing reads), ragged last tile."""
    torch.manual_seed(0)
    (n, N), K = shape, 75
    w = aq.ActorCriticWeights.random(hidden, K, 4, seed=3)
    w.std = 0.1
    obs = torch.randn(n, K, N, device="cuda")
    act_net, cri_net = _torch_nets(w)
    with torch.no_grad():
        mean0 = act_net(obs.permute(0, 2, 1).double())  # [n, N, 4]
    actions = (mean0 + 0.1 * torch.randn_like(mean0)).permute(0, 2, 1).float().contiguous()      # [n, 4, N]
    old_lp = torch.distributions.Normal(mean0 + 0.03 * torch.randn_like(mean0), 0.1).log_prob(actions.permute(0, 2, 1).double())
    old_lp = old_lp.permute(0, 2, 1).float().contiguous()
    adv = torch.randn(n, N, device="cuda"); rets = 2 * torch.randn(n, N, device="cuda")
    learner = aq.PpoLearner(w, K_epochs=1)
    assert learner.fused_loss
    learner.fused_loss = fused
    loss = learner.gradients(obs, actions, old_lp, adv, rets)
    assert abs(float(loss["loss"]) - float(loss["actor"] + loss["critic"] + loss["entropy"])) < 1e-12
    # torch reference of the same loss
    x = obs.permute(0, 2, 1).double()
    mean = act_net(x); value = cri_net(x).squeeze(-1)
    dist = torch.distributions.Normal(mean, 0.1)
    lp = dist.log_prob(actions.permute(0, 2, 1).double()).sum(-1)
    ratio = torch.exp(lp - old_lp.permute(0, 2, 1).double().sum(-1))
    a_ = adv.double()
    total = (-torch.min(ratio * a_, torch.clamp(ratio, 0.8, 1.2) * a_)).mean() + 0.5 * ((value - rets.double()) ** 2).mean() \
        - 0.006 * dist.entropy().sum(-1).mean()
    total.backward()
    assert abs(float(loss["loss"]) - float(total.detach())) < 1e-4 * max(1.0, abs(float(total.detach())))
    ref_grads = {"actor": [act_net[0].weight.grad, act_net[0].bias.grad, act_net[2].weight.grad, act_net[2].bias.grad,
                           act_net[4].weight.grad, act_net[4].bias.grad],
                 "critic": [cri_net[0].weight.grad, cri_net[0].bias.grad, cri_net[2].weight.grad, cri_net[2].bias.grad,
                            cri_net[4].weight.grad, cri_net[4].bias.grad]}
    for head in ("actor", "critic"):
        for k in range(6):
            g = learner._state[(head, k)][0].double()
            r = ref_grads[head][k]
            scale = float(r.abs().max()) + 1e-12
            assert float((g - r).abs().max()) / scale < 2e-4, (head, k, float((g - r).abs().max()), scale)
    # one Adam step vs torch.optim.Adam (fp32 parameters)
    act32, cri32 = _torch_nets(w, torch.float32)
    opt = torch.optim.Adam(list(act32.parameters()) + list(cri32.parameters()), lr=5e-4, betas=(0.9, 0.999))
    for net, head in ((act32, "actor"), (cri32, "critic")):
        for lin, (kw, kb) in zip([net[0], net[2], net[4]], ((0, 1), (2, 3), (4, 5))):
            lin.weight.grad = learner._state[(head, kw)][0].clone(); lin.bias.grad = learner._state[(head, kb)][0].clone()
    opt.step()
    learner.adam_step()
    for net, head in ((act32, "actor"), (cri32, "critic")):
        for lin, (kw, kb) in zip([net[0], net[2], net[4]], ((0, 1), (2, 3), (4, 5))):
            assert torch.allclose(w.head(head)[kw], lin.weight, rtol=0, atol=2e-7)
            assert torch.allclose(w.head(head)[kb], lin.bias, rtol=0, atol=2e-7)


HANDLES = {"lean": dict(dtype=torch.float64, compute="lean"), "f32": dict(dtype=torch.float32)}


@pytest.mark.parametrize("handle", ["lean", "f32"])
@pytest.mark.parametrize("policy", ["random", "trained"])
def test_full_size_ppo_rollout_properties_config5(aq, handle, policy):
    """BASELINE configs[4] at its per-GPU size (65 536 envs, T=5, N=128 actor + critic, Normal(mean, 0.1) Philox
    sampling, masked auto-reset, GAE): size-independent properties instead of an oracle that would need hours --
    (i) replay: the same seed gives the same bits; (ii) shard invariance: a 16 384-env shard with its env_offset
    reproduces rows [32768, 49152) of the full batch bit for bit (actions, rewards, masks, values); (iii) the GAE
    identity returns - raw advantage = V and the reverse-scan recurrence, checked on device tensors; (iv) masks =
    1 - done and the episode counter equals the number of zeros in the masks."""
    N, S = 65536, 24
    # bench.py's two policies: the reference's trained N=128 weights (reset-light) and random-init ones (reset-heavy), on
    # the handle bench.py measures on (lean) and on fp32
    w = _weights(aq)[0] if policy == "trained" else aq.ActorCriticWeights.random(128, 75, 4, seed=1)
    w.std = 0.1

    def run(n, lo):
        # random policy: 40-step episodes, the resets come from bounding-box exits; trained policy: 12-step episodes, so
        # that every env also passes through a time-out reset inside the 24 steps
        env = aq.BatchedQuad(n, 0.01, 40 if policy == "random" else 12, training=True, direct_control=1, T=5, **HANDLES[handle])
        ro = aq.PolicyRollout(env, w, S, auto_reset=True, seed=5, env_offset=lo)
        ro.start()
        for _ in range(S):
            ro.step()
        return env, ro
    env, ro = run(N, 0)
    env2, ro2 = run(N, 0)
    for a, b in ((ro.actions, ro2.actions), (ro.rewards, ro2.rewards), (ro.masks, ro2.masks), (ro.values, ro2.values),
                 (env.state, env2.state)):
        assert torch.equal(a, b)
    envs, ros = run(16384, 32768)
    sl = slice(32768, 49152)
    assert torch.equal(ros.actions, ro.actions[:, :, sl]) and torch.equal(ros.rewards, ro.rewards[:, sl])
    assert torch.equal(ros.masks, ro.masks[:, sl]) and torch.equal(ros.values[:S], ro.values[:S, sl])
    assert torch.equal(envs.state, env.state[:, sl]) and torch.equal(envs.episodes, env.episodes[sl])
    assert bool(torch.isfinite(ro.rewards).all()) and float(ro.actions.abs().max()) < 2.0
    # GAE identities on the un-normalised quantities
    from autonomous_quadrotor_environment_amd import gae
    vals = ro.values.clone(); vals[S] = 0
    ret, adv = gae(vals, ro.masks, ro.rewards, ddof=0, normalize=False, group=False)
    assert torch.allclose(ret - adv, vals[:S], atol=1e-5)
    delta = ro.rewards + 0.99 * vals[1:] * ro.masks - vals[:S]
    g_next = torch.cat([adv[1:], torch.zeros(1, N, device="cuda", dtype=adv.dtype)])
    assert torch.allclose(adv, delta + 0.99 * 0.99 * ro.masks * g_next, atol=2e-3, rtol=1e-5)
    assert int((ro.masks == 0).sum()) == int(env.episodes.sum())
    # episodes end through the bounding box (random policy: thousands at this size), the 40-step limit, or the solved box
    assert int(env.episodes.sum()) > 1000


@pytest.mark.parametrize("handle", ["lean", "f32"])
def test_full_size_actor_in_loop_is_batch_position_independent_config4(aq, handle):
    """BASELINE configs[3] at its per-GPU size (131 072 envs, deterministic actor in the loop): the trajectory of an
    env does not depend on how many other envs share its launch (tiles, grid-stride order, workgroup) -- the first
    4 096 envs alone reproduce rows [0, 4096) of the full batch bit for bit over 20 steps."""
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    N, n = 131072, 4096
    w = _weights(aq)[0]  # config 4's actor: the reference's checked-in N=128 weights
    x0 = torch.from_numpy(np.ascontiguousarray(fast_random_init_states(4, N).T)).to("cuda", HANDLES[handle]["dtype"])

    def run(cnt):
        env = aq.BatchedQuad(cnt, 0.01, 1000, training=False, direct_control=1, T=5, **HANDLES[handle])
        ro = aq.PolicyRollout(env, aq.ActorCriticWeights(w.actor, None), 20)
        ro.start(x0[:, :cnt].contiguous())
        for _ in range(20):
            ro.step(deterministic=True)
        return env, ro
    big, rb = run(N)
    small, rs = run(n)
    assert torch.equal(rs.actions, rb.actions[:, :, :n]) and torch.equal(small.state, big.state[:, :n])
    assert float(rb.actions.abs().max()) <= 1.0 and bool(torch.isfinite(big.state).all())


def test_ppo_update_against_the_reference_update(aq):
    """PPO.update of the reference run by its own classes (g21: 96 collected samples, three episode ends, hidden 16,
    K_epochs = 10, Adam lr 5e-4): GAE on the device reproduces its returns / normalised advantages, and ten epochs of
    PpoLearner (fp32 kernels) land on the reference's fp64 weights.  Adam's first steps move every weight by ~lr
    whatever the gradient's size, so a weight whose gradient is within fp32 noise of zero may differ by a few lr; the
    bulk must agree to 1e-5."""
    import types
    g = golden("g21_ppo_update")
    w = aq.ActorCriticWeights.from_state_dict({k[3:]: g[k] for k in g.files if k.startswith("w0_") and k != "w0_std"})
    w.std = float(g["w0_std"])
    B = g["obs"].shape[0]
    dev = "cuda"
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, torch.float32)
    values = f(g["values"]).reshape(B + 1, 1)
    masks = f(1.0 - g["is_terminals"]).reshape(B, 1)
    rewards = f(g["rewards"]).reshape(B, 1)
    ret, adv = aq.gae(values, masks, rewards, ddof=0, group=False)
    assert rel_err(_np(ret).ravel(), g["returns"], 1.0) < 2e-6
    assert rel_err(_np(adv).ravel(), g["adv"], 1.0) < 2e-5
    ro = types.SimpleNamespace(t=B, obs=f(g["obs"]).reshape(B, 75, 1), actions=f(g["actions"]).reshape(B, 4, 1),
                               logprobs=f(g["logprobs"]).reshape(B, 4, 1))
    learner = aq.PpoLearner(w, lr=5e-4, K_epochs=10, group=False)
    learner.update(ro, ret, adv)
    names = [f"{h}_{l}_{p}" for h in ("actor", "critic") for l in (0, 2, 4) for p in ("weight", "bias")]
    got = np.concatenate([_np(t).ravel() for h in ("actor", "critic") for t in w.head(h)])
    ref = np.concatenate([g["w1_" + n].ravel() for n in names])
    start = np.concatenate([g["w0_" + n].ravel() for n in names])
    d = np.abs(got - ref)
    moved = np.abs(ref - start)
    assert moved.max() > 4e-3                      # the update did move the weights by ~K lr
    assert np.mean(d < 1e-5) > 0.97, np.mean(d < 1e-5)
    assert d.max() < 2.5e-3, d.max()               # worst case: a near-zero gradient component flipping sign for some epochs
    assert np.median(d) < 2e-6


def test_landing_trainer_update_against_the_reference_update(aq):
    """a22's second variant pinned by the reference itself (fixture g26 = visual_landing/ppo_trainer.py's OWN optimizer_step /
    update loop, :96-131,:187-196, run on an MLP-backed subclass of its rl_nn_model_v2.ActorCritic): K = 5 epochs over the
    reference's recorded permutations in minibatches of 1536 + the remainder of 500 = 15 Adam steps (lr 2e-5), loss
    actor.mean() + 0.01 (-H).mean() + 0.5 MSE on the JOINT action probability, the std trained with the weights.
    PpoLearner(c_entropy=0.01, learn_std=True).update(minibatch=1536, perms=...) lands on the reference's weights: they moved
    by up to 3e-4 (15 steps of ~lr each), the device is within 2e-6 of them on 97 % (fp32 kernels against fp32 torch; a
    weight whose gradient sits at fp32 noise can differ by a few lr, as in the g21 test), and the trained std agrees to 1e-6."""
    import types
    g = golden("g26_landing_trainer_update")
    w = aq.ActorCriticWeights.from_state_dict({k[3:]: g[k] for k in g.files if k.startswith("w0_") and k != "w0_std"})
    w.std = float(g["w0_std"])
    n = g["sens"].shape[0]
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch.float32)
    # get_advantages of the trainer (ddof 1) on the device reproduces what update() derived (float32 recurrences there)
    ret, adv = aq.gae(f(np.append(g["values"], 0.0)).reshape(n + 1, 1), f(1.0 - g["is_terminals"]).reshape(n, 1),
                      f(g["rewards"]).reshape(n, 1), ddof=1, group=False)
    assert rel_err(_np(ret).ravel(), g["returns"], 1.0) < 2e-5 and rel_err(_np(adv).ravel(), g["adv"], 1.0) < 5e-5
    # the optimisation itself on the reference's own returns / advantages (so that the comparison below is the update's)
    lp = np.zeros((n, 4)); lp[:, 0] = g["logprobs"]      # the trainer stores the JOINT log-probability: the kernel sums the rows
    ro = types.SimpleNamespace(t=n, obs=f(g["sens"]).reshape(n, 75, 1), actions=f(g["actions"]).reshape(n, 4, 1),
                               logprobs=f(lp).reshape(n, 4, 1))
    learner = aq.PpoLearner(w, lr=float(g["lr"]), K_epochs=int(g["K_epochs"]), eps_clip=float(g["eps_clip"]),
                            c_value=float(g["c_value"]), c_entropy=float(g["c_entropy"]), group=False, learn_std=True)
    out = learner.update(ro, f(g["returns"]).reshape(n, 1), f(g["adv"]).reshape(n, 1), minibatch=int(g["batch_size"]),
                         perms=g["perms"])
    assert len(out) == 15 and learner.step == 15
    assert abs(float(np.mean([float(o["actor"]) for o in out])) - float(g["actor_loss_mean"])) < 2e-6
    assert abs(float(np.mean([float(o["critic"]) for o in out])) - float(g["critic_loss_mean"])) < 2e-5
    names = [f"{h}_{l}_{p}" for h in ("actor", "critic") for l in (0, 2, 4) for p in ("weight", "bias")]
    got = np.concatenate([_np(t).ravel() for h in ("actor", "critic") for t in w.head(h)])
    ref = np.concatenate([g["w1_" + k].ravel() for k in names])
    start = np.concatenate([g["w0_" + k].ravel() for k in names])
    d, moved = np.abs(got - ref), np.abs(ref - start)
    assert moved.max() > 2.5e-4
    assert np.mean(d < 2e-6) > 0.97 and np.median(d) < 2e-7 and d.max() < 1e-4, (np.mean(d < 2e-6), np.median(d), d.max())
    assert abs(w.std - float(g["w1_std"])) < 1e-6, (w.std, float(g["w1_std"]))
    with pytest.raises(ValueError):
        learner.update(ro, f(g["returns"]).reshape(n, 1), f(g["adv"]).reshape(n, 1), minibatch=1536, perms=g["perms"][:2])


def test_ppo_update_from_a_hip_graph_equals_the_eager_update(aq):
    """PpoLearner.update(graph=True): one epoch captured as a hipGraph (Adam's step count in device memory) and replayed
    K_epochs times, two updates in a row -- the same weights, bit for bit, as the eager launches (deterministic sums)."""
    import types
    torch.manual_seed(3)
    n, N = 16, 64
    ro = types.SimpleNamespace(t=n, obs=torch.randn(n, 75, N, device="cuda") * 0.3, actions=torch.randn(n, 4, N, device="cuda") * 0.3,
                               logprobs=torch.randn(n, 4, N, device="cuda") * 0.1 + 1.3)
    adv, ret = torch.randn(n, N, device="cuda"), torch.randn(n, N, device="cuda")
    got = {}
    for graph in (False, True):
        w = aq.ActorCriticWeights.random(32, seed=5)
        learner = aq.PpoLearner(w, K_epochs=4, group=False, deterministic=True)
        l1 = learner.update(ro, ret, adv, graph=graph)
        l2 = learner.update(ro, ret, adv, graph=graph)
        assert learner.step == 8 and len(l1) == 4 and len(l2) == 4
        got[graph] = (learner.flat_p.clone(), torch.stack([x["loss"] for x in l1 + l2]).clone())
    assert torch.equal(got[True][0], got[False][0])
    assert torch.allclose(got[True][1], got[False][1], rtol=0, atol=1e-12)  # the reported loss sums are double atomics
    assert float((got[True][1][0] - got[True][1][-1]).abs()) > 1e-3  # the loss did move over the eight steps


def test_ppo_update_with_learned_std_against_the_reference(aq):
    """g23 = the reference's PPO.update with fixed_std = False (model.py:45-48: std is an nn.Parameter with grad, ppo.py:105
    hands it to Adam with the weights): PpoLearner(learn_std=True) moves the action std to the reference's value (0.1 ->
    0.095643 in ten epochs) and the weights with it."""
    import types
    g = golden("g23_ppo_update_learned_std")
    assert abs(float(g["w1_std"]) - float(g["w0_std"])) > 3e-3  # the reference did train it
    w = aq.ActorCriticWeights.from_state_dict({k[3:]: g[k] for k in g.files if k.startswith("w0_") and k != "w0_std"})
    w.std = float(g["w0_std"])
    B = g["obs"].shape[0]
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch.float32)
    ret, adv = aq.gae(f(g["values"]).reshape(B + 1, 1), f(1.0 - g["is_terminals"]).reshape(B, 1), f(g["rewards"]).reshape(B, 1),
                      ddof=0, group=False)
    ro = types.SimpleNamespace(t=B, obs=f(g["obs"]).reshape(B, 75, 1), actions=f(g["actions"]).reshape(B, 4, 1),
                               logprobs=f(g["logprobs"]).reshape(B, 4, 1))
    learner = aq.PpoLearner(w, lr=5e-4, K_epochs=10, group=False, learn_std=True)
    learner.update(ro, ret, adv)
    assert abs(w.std - float(g["w1_std"])) < 2e-6, (w.std, float(g["w1_std"]))
    names = [f"{h}_{l}_{p}" for h in ("actor", "critic") for l in (0, 2, 4) for p in ("weight", "bias")]
    got = np.concatenate([_np(t).ravel() for h in ("actor", "critic") for t in w.head(h)])
    ref = np.concatenate([g["w1_" + n].ravel() for n in names])
    d = np.abs(got - ref)
    assert np.mean(d < 1e-5) > 0.97 and np.median(d) < 2e-6 and d.max() < 2.5e-3
    with pytest.raises(ValueError):
        aq.PpoLearner(aq.ActorCriticWeights.random(256), learn_std=True)


@pytest.mark.parametrize("hidden", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("T,N", [(5, 4097), (5, 63), (3, 1000), (1, 1)])
def test_mlp_forward_every_instantiation_vs_torch_fp64(aq, hidden, T, N):
    """Every k_mlp instantiation (hidden 16..256; T = 5 takes the unrolled window path, other T the run-time one; ragged
    and single-env batches; W2 streamed from L2 at hidden 256) against a torch fp64 MLP on the same window, actor
    (tanh head) and critic, after the ring head has moved.  Tolerance 2e-6 abs: fp32 MFMA chains + the 1.2e-7 tanh."""
    torch.manual_seed(hidden * 7 + T)
    w = aq.ActorCriticWeights.random(hidden, 15 * T, 4, seed=hidden + T)
    env = aq.BatchedQuad(N, 0.01, 1000, training=False, direct_control=1, T=T)
    env.reset(aq.random_init_states(11, N).T.copy(), return_history=False, push_window=True)
    for _ in range(T + 2):  # moves the ring head off slot 0
        env.step(torch.randn(4, N, device="cuda").clamp(-1, 1), push_window=True)
    obs = torch.empty(15 * T, N, device="cuda")
    mean = env.mlp_forward(w, "actor", obs_store=obs)
    val = env.mlp_forward(w, "critic")
    act_net, cri_net = _torch_nets(w)
    x = obs.t().double()
    with torch.no_grad():
        ref_m, ref_v = act_net(x).t(), cri_net(x).t()
    assert float((mean.double() - ref_m).abs().max()) < 2e-6
    assert float((val.double() - ref_v).abs().max()) < 5e-6 * max(1.0, float(ref_v.abs().max()))


def test_minibatch_schedule_of_the_landing_trainer(aq):
    """ppo_trainer.py:187-196 on the MLP policy: K epochs x ceil(n / per) Adam steps over fresh permutations of the
    stored steps; with one batch covering everything it degenerates to the full-batch update (same weights up to the
    fp32 atomics' summation order), and the loss of the clipped objective falls over the epochs."""
    torch.manual_seed(3)
    w = aq.ActorCriticWeights.random(32, 75, 4, seed=5); w.std = 0.1
    env = aq.BatchedQuad(64, 0.01, 200, training=True, direct_control=1, T=5)
    ro = aq.PolicyRollout(env, w, 24, store_obs=True, auto_reset=True, seed=9)
    ro.start()
    for _ in range(24):
        ro.step()
    ret, adv = ro.finish(ddof=1, group=False)   # the trainer's ddof (ppo_trainer.py:148)
    import copy
    w2 = aq.ActorCriticWeights([t.clone() for t in w.actor], [t.clone() for t in w.critic]); w2.std = 0.1
    full = aq.PpoLearner(w, lr=2e-5, K_epochs=5, c_entropy=0.01, group=False)
    mini = aq.PpoLearner(w2, lr=2e-5, K_epochs=5, c_entropy=0.01, group=False)
    a = full.update(ro, ret, adv)
    b = mini.update(ro, ret, adv, minibatch=64 * 24)           # one batch = everything
    assert len(a) == 5 and len(b) == 5
    for x, y in zip(w.actor + w.critic, w2.actor + w2.critic):
        assert torch.allclose(x, y, rtol=0, atol=5e-6)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    c = mini.update(ro, ret, adv, minibatch=64 * 5, generator=g)  # 24 steps in batches of 5 -> 5 batches (4 x 5 + 4)
    assert len(c) == 5 * 5 and all(np.isfinite(float(l["loss"])) for l in c)
    assert mini.step == 5 + 25


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("hidden", [32, 128])
def test_batched_critic_equals_per_step_critic(aq, dt, hidden):
    """V(s_t) of the whole horizon in one launch over the stored observations at finish() (the default once the
    observations are stored) against one critic launch per step: same values, returns and raw advantages, bitwise."""
    w = aq.ActorCriticWeights.random(hidden, 75, 4, seed=3)
    n, H = 777, 37
    outs = []
    for batched in (True, False):
        env = aq.BatchedQuad(n, 0.01, 15, training=True, direct_control=1, T=5, dtype=dt)
        ro = aq.PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=2)
        assert ro.batched_critic
        ro.batched_critic = batched
        ro.start()
        for _ in range(H):
            ro.step()
        if batched:
            assert float(ro.values.abs().max()) == 0.0  # nothing evaluated during the rollout
        ret, adv = ro.finish(ddof=0, group=False, bootstrap=True)
        outs.append((ro.values.clone(), ret, adv, ro.obs.clone()))
    assert float(outs[0][0].abs().max()) > 0
    assert torch.equal(outs[0][3], outs[1][3]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][2], outs[1][2], rtol=0, atol=1e-5)  # normalisation moments: fp64 atomics
    # no stored observations / another window length: one launch per step
    env = aq.BatchedQuad(64, 0.01, 15, training=True, direct_control=1, T=5)
    assert not aq.PolicyRollout(env, w, 4).batched_critic


@pytest.mark.parametrize("N", [100, 40000])
def test_prepared_weight_image_equals_in_kernel_staging(aq, N):
    """qenv_mlp_prepare (the weights in the forward kernel's LDS layout, built once per weight update) against the same
    kernels staging the fp32 weights themselves: bit-identical actor and critic outputs, for the tile-per-workgroup kernel
    (100 envs) and the staged one (40 000); after an in-place weight change + touch() the image follows."""
    from autonomous_quadrotor_environment_amd import _lib as L
    w, _ = _weights(aq, 128)
    env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5)
    env.reset(env.sample_init_states(3, 0, 0), return_history=False, push_window=True)

    def both():
        return env.mlp_forward(w, "actor").clone(), env.mlp_forward(w, "critic").clone()

    def unregistered():
        for slot in (0, 1):
            L.check(L.lib.qenv_mlp_prepare(env._h, slot, 75, 128, 4, None, None, None, None, None, None, None), "unregister")
        keep, env._prep = env._prep, {}
        ensure, env._ensure_prepared = env._ensure_prepared, (lambda *a, **k: None)
        try:
            return both()
        finally:
            env._prep, env._ensure_prepared = keep, ensure
            for slot, head in ((0, "actor"), (1, "critic")):   # re-register the images that were built
                W = w.head(head)
                L.check(L.lib.qenv_mlp_prepare(env._h, slot, 75, 128, int(W[4].shape[0]), *[L.ptr(t) for t in W],
                                               L.ptr(env._prep[slot][1])), "re-register")
    a1, c1 = both()
    assert set(env._prep) == {0, 1}
    a0, c0 = unregistered()
    assert torch.equal(a1, a0) and torch.equal(c1, c0)
    w.actor[0].mul_(1.01); w.critic[4].add_(0.25); w.touch()
    a2, c2 = both()
    assert not torch.equal(a2, a1) and not torch.equal(c2, c1)
    a3, c3 = unregistered()
    assert torch.equal(a2, a3) and torch.equal(c2, c3)


def test_an_alias_of_the_trained_weights_follows_the_learner(aq):
    """A persistent evaluation env that holds `ActorCriticWeights(w.actor, None)` -- a second wrapper of the tensors the
    learner trains -- must run the NEW policy after an Adam step: the version that keys the env's prepared weight image
    belongs to the storage, not to the wrapper.  Checked against a fresh env (fresh image) on the same window, bit for
    bit, for the wrapper made by hand and by actor_only()."""
    w = aq.ActorCriticWeights.random(128, 75, 4, seed=4)
    learner = aq.PpoLearner(w, K_epochs=1)
    aliases = [aq.ActorCriticWeights(w.actor, None), w.actor_only()]
    assert all(a.actor[0].data_ptr() == w.actor[0].data_ptr() for a in aliases)
    init = None
    envs = []
    for a in aliases:
        ev = aq.BatchedQuad(96, 0.01, 500, training=False, direct_control=1, T=5)
        init = ev.sample_init_states(5, 0, 0) if init is None else init
        ev.reset(init.clone(), return_history=False, push_window=True)
        envs.append(ev)
    before = [ev.mlp_forward(a, "actor").clone() for ev, a in zip(envs, aliases)]
    assert torch.equal(before[0], before[1])
    learner.bucket.flat.normal_(generator=torch.Generator(device="cuda").manual_seed(1))   # any gradient
    learner.adam_step()
    after = [ev.mlp_forward(a, "actor").clone() for ev, a in zip(envs, aliases)]
    fresh = aq.BatchedQuad(96, 0.01, 500, training=False, direct_control=1, T=5)
    fresh.reset(init.clone(), return_history=False, push_window=True)
    want = fresh.mlp_forward(w, "actor")
    assert not torch.equal(after[0], before[0])
    assert torch.equal(after[0], want) and torch.equal(after[1], want)
    # a wrapper that COPIES (fp64 numbers in) is its own storage with its own version
    copy = aq.ActorCriticWeights([t.double() for t in w.actor], None)
    assert copy.actor[0].data_ptr() != w.actor[0].data_ptr() and copy.version_of("actor") != w.version_of("actor")


@pytest.mark.parametrize("hidden", [64, 128])
@pytest.mark.parametrize("n_out,final_tanh", [(4, True), (1, False)])
def test_small_batch_mlp_kernel_is_bit_identical_to_the_staged_kernel(aq, hidden, n_out, final_tanh):
    """Up to 512 tiles of 32 envs the forward runs on k_mlp_small (a tile split over HP/32 waves, weights straight from
    L2, activations exchanged through LDS), beyond that on the LDS-staged k_mlp: the same observations must give the
    same bits whichever kernel their batch size selects (k-pairs and accumulation order are kept)."""
    g = torch.Generator(device="cuda").manual_seed(hidden + n_out)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    W = [r(hidden, 75) * 0.2, r(hidden) * 0.1, r(hidden, hidden) * 0.15, r(hidden) * 0.1, r(n_out, hidden) * 0.2, r(n_out) * 0.1]
    x = r(75, 40000)
    big = aq.dense_tail(x, W, final_tanh=final_tanh)                      # 1250 tiles: k_mlp
    for n in (1, 31, 1000, 16384):                                       # <= 512 tiles: k_mlp_small
        small = aq.dense_tail(x[:, :n].contiguous(), W, final_tanh=final_tanh)
        assert torch.equal(small, big[:, :n]), n
    ref = torch.tanh(W[2].double() @ torch.tanh(W[0].double() @ x.double() + W[1].double()[:, None]) + W[3].double()[:, None])
    ref = W[4].double() @ ref + W[5].double()[:, None]
    ref = torch.tanh(ref) if final_tanh else ref
    assert float((big.double() - ref).abs().max()) < 5e-6


def test_episode_return_is_the_workers_reward_sum(aq):
    """ppo.py:236,258: reward_sum is zeroed AFTER env.reset() and sums the rewards env.step returned to the loop -- the
    rewards of the reset's own T warm-up steps are not part of it (found by tests/tools/fuzz_policy_rollout.py)."""
    n, T = 200, 5
    env = aq.BatchedQuad(n, 0.01, 10, training=True, direct_control=1, T=T, dtype=torch.float64)
    env.reset(env.sample_init_states(1, 0, 0), push_window=True)
    assert int(env.i.min()) == T and float(env.reward.abs().max()) > 0      # warm-up steps did earn rewards ...
    assert float(env.episode_return.abs().max()) == 0.0                     # ... which the episode does not count
    total = torch.zeros(n, device="cuda", dtype=torch.float64)
    act = torch.zeros(4, n, device="cuda", dtype=torch.float64)
    for k in range(4):
        _, r, d = env.step(act, push_window=True)
        total += r
        assert torch.equal(env.episode_return, total)
        env.reset_sampled(1, k + 1, 0, mask=d, push_window=True)
        assert torch.equal(env.last_return[d], total[d]) and float(env.episode_return[d].abs().max() if bool(d.any()) else 0.0) == 0.0
        total[d] = 0


def test_deterministic_training_is_bit_reproducible(aq):
    """PpoLearner(deterministic=True) + finish(deterministic=True): weight / bias gradients and the advantage moments are
    summed in a fixed order instead of with atomics -- two runs of rollout -> GAE -> K epochs end with identical
    weights, and the deterministic gradients agree with the atomic ones to summation-order noise."""
    def run(det):
        w = aq.ActorCriticWeights.random(64, 75, 4, seed=9)
        env = aq.BatchedQuad(1000, 0.01, 60, training=True, direct_control=1, T=5)
        ro = aq.PolicyRollout(env, w, 20, store_obs=True, auto_reset=True, seed=3)
        ln = aq.PpoLearner(w, K_epochs=3, deterministic=det)
        ro.start()
        for _ in range(3):
            ro.t = 0
            for _ in range(20):
                ro.step()
            ret, adv = ro.finish(bootstrap=True, deterministic=det)
            ln.update(ro, ret, adv)
        return [t.clone() for t in w.actor + w.critic], [ln._state[k][0].clone() for k in ln._state]
    w1, g1 = run(True)
    w2, g2 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(w1, w2)) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    w3, g3 = run(False)
    for a, b in zip(w1, w3):
        assert float((a - b).abs().max()) < 1e-3  # three Adam steps of K = 3 epochs apart by atomics noise only


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("hidden,N,T", [(128, 1000, 5), (128, 40000, 5), (64, 777, 5), (32, 300, 5), (128, 500, 3), (256, 200, 5)])
def test_policy_act_equals_forward_then_sample(aq, dt, hidden, N, T):
    """qenv_policy_act (ActorCritic.act in one launch: the sampling runs in the epilogue of the actor forward) against
    qenv_mlp_forward + qenv_sample_action: means, actions, log-probs and the stored observation, bit for bit, with the
    Philox streams and with injected noise, across the kernel variants (k_mlp_small, k_mlp, run-time T, hidden 256)."""
    w = aq.ActorCriticWeights.random(hidden, 15 * T, 4, seed=4)
    w.std = 0.3
    env = aq.BatchedQuad(N, 0.01, 1000, training=False, direct_control=1, T=T, dtype=dt)
    env.reset(env.sample_init_states(2, 0, 0), return_history=False, push_window=True)
    env.step(torch.zeros(4, N, device="cuda", dtype=dt), push_window=True)  # ring head off zero
    for noise in (None, torch.randn(4, N, device="cuda", dtype=dt)):
        mean = env.mlp_forward(w, "actor")
        lp = torch.empty(4, N, device="cuda", dtype=dt)
        act = env.sample_action(mean, w.std, noise, seed=9, step=17, env_offset=123, logprob=lp)
        lp2, mean2 = torch.empty_like(lp), torch.empty_like(mean)
        obs = torch.empty(15 * T, N, device="cuda")
        act2 = env.policy_act(w, noise, seed=9, step=17, env_offset=123, logprob=lp2, mean=mean2, obs_store=obs)
        assert torch.equal(mean, mean2) and torch.equal(act, act2) and torch.equal(lp, lp2)
        obs_ref = torch.empty_like(obs)
        env.mlp_forward(w, "actor", obs_store=obs_ref)
        assert torch.equal(obs, obs_ref)
        assert float((act - mean).abs().max()) > 0.1


def test_bench_gpus2_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts the two ranks itself (gloo here: one
    GPU box), rank 0's line reports n_gpus = 2, twice the per-GPU batch, and the all-gathered episode returns have
    the global length (ppo.py:356-382's Pool gather)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["QENV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "config3", "--envs", "8192",
                        "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--min-seconds", "0.05"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 8192 and d["config"]["global_envs"] == 2 * 8192
    assert d["config"]["all_gathered_returns"] == 2 * 8192 and d["config"]["finite"]
    assert d["value"] > 0 and d["timing"]["regions"] >= 1
    assert d["dist_backend"] == "gloo" and d["rccl_ranks"] == 0   # under nccl: rccl_ranks == n_gpus


def test_bench_gpus2_default_line_carries_config4_and_config5_with_the_all_gather():
    """The default workload under --gpus 2 (what a SCALE run launches, gloo here): besides config 3 (the headline) the line
    carries BASELINE's two node-level configs at their per-GPU sizes -- config 4 (trained actor in the loop) and config 5
    (full PPO rollout + GAE) -- each ending in the all-gather of the episode returns with the GLOBAL length, and the
    gather's time; the whole slices are in the detail line, which the launcher parent relays in front of the contract line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["QENV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--min-seconds", "0.05", "--no-sustained"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines[-1].startswith('{"metric"') and lines[0].startswith('{"bench_detail"') and len(lines) == 2
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * 262144 and d["config"]["handle"] == "lean"
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and d["config"]["all_gathered_returns"] == 2 * 262144
    full = json.loads(lines[0])["bench_detail"]["configs"]
    assert full["config4"]["dominant_kernel"].startswith("k_ppo_fused") and full["config5"]["schedule"] == "inband"
    c4, c5 = d["configs"]["config4"], d["configs"]["config5"]
    assert c4["global_envs"] == 2 * 131072 and c4["all_gathered_returns"] == 2 * 131072 and c4["finite"]
    assert c5["global_envs"] == 2 * 65536 and c5["all_gathered_returns"] == 2 * 65536 and c5["finite"]
    assert c4["weights"].startswith("trained") and c5["weights"].startswith("trained")
    assert c4["all_gather_us"] > 0 and c5["all_gather_us"] > 0 and c4["value"] > 0 and c5["value"] > 0


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_ppo_step_fused_call_is_bit_identical(overlap):
    """qenv_ppo_step (reset in the step kernel's epilogue; QENV_PPO_OVERLAP=1: warm-up on a side stream beside the next
    policy step of every other env) against the separate policy_act / step / reset_sampled calls: every rollout buffer,
    the GAE outputs, the env state after finish(), banked returns and episode counts equal bit for bit over three
    horizons (tests/tools/ab_ppo_overlap.py; the variable is read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n, h, handle in ((4000, 24, "f32"), (333, 24, "lean")):
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "ab_ppo_overlap.py"), str(n), str(h), handle],
                           env=dict(os.environ, QENV_PPO_OVERLAP=overlap), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
        assert '"mismatches": []' in r.stdout


@pytest.mark.gpu
def test_weights_of_another_window_are_refused(aq):
    """A weight set built for T = 5 (75 inputs) on a T = 3 handle would be mis-strided by the kernels: the host mirror
    refuses it by shape, and the C-ABI itself by k_in (the row length of W1 the caller passes)."""
    import torch
    from autonomous_quadrotor_environment_amd import _lib as L
    env = aq.BatchedQuad(64, 0.01, 100, training=True, direct_control=1, T=3)
    env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
    w75 = aq.ActorCriticWeights.random(64, 75, 4, device="cuda", seed=0)
    with pytest.raises(ValueError, match="T=3"):
        env.mlp_forward(w75, "actor")
    with pytest.raises(ValueError, match="T=3"):
        env.policy_act(w75)
    W = w75.head("actor")
    out = torch.empty(4, 64, device="cuda")
    env._bind()
    rc = L.lib.qenv_mlp_forward(env._h, 75, 64, 4, 1, *[L.ptr(t) for t in W], L.ptr(out), None)
    assert rc == -1 and b"k_in" in L.lib.qenv_last_error()
    rc = L.lib.qenv_policy_act(env._h, 75, 64, *[L.ptr(t) for t in W], 0.1, None, 0, 0, 0, None, L.ptr(out), None, None)
    assert rc == -1 and b"k_in" in L.lib.qenv_last_error()
    w45 = aq.ActorCriticWeights.random(64, 45, 4, device="cuda", seed=0)
    assert torch.isfinite(env.mlp_forward(w45, "actor")).all()


@pytest.mark.gpu
@pytest.mark.parametrize("inband", [False, True])
@pytest.mark.parametrize("shards", [2, 3])
def test_stream_sharded_rollout_is_the_single_batch_rollout(aq, shards, inband):
    """StreamShardedRollout (the GPU's env range as S shards on S streams): every rollout buffer of the union equals the
    single-batch PolicyRollout bit for bit (trajectories, Philox streams and resets are keyed by the global env index),
    returns too; the advantages are normalised with the moments of ALL shards (summed in another order: 1e-12).
    inband: the in-band schedule with the observations stored as slices (the configuration bench.py's config 5 runs; the
    second horizon starts from slot 0 again and must store its whole first window)."""
    import torch
    N, H = 1000, 24
    kw = dict(inband=True, key_by_episode=True) if inband else {}
    w = aq.ActorCriticWeights.random(128, 75, 4, device="cuda", seed=2); w.std = 0.3

    def make_env(lo, hi):
        return aq.BatchedQuad(hi - lo, 0.01, 40, training=True, direct_control=1, T=5, dtype=torch.float64, compute="lean")
    one = aq.PolicyRollout(make_env(0, N), w, H, store_obs=True, auto_reset=True, seed=9, env_offset=100, **kw)
    one.start()
    sh = aq.StreamShardedRollout(make_env, N, w, H, shards=shards, seed=9, env_offset=100, store_obs=True, auto_reset=True, **kw)
    sh.start()
    assert one.slice_obs == inband and all(r.slice_obs == inband for r in sh.ros)
    for rep in range(2):   # the second horizon starts with resets pending from the first
        one.t = 0; sh.t = 0
        for _ in range(H):
            one.step(); sh.step()
        ret1, adv1 = one.finish(ddof=0, group=False)
        outs = sh.finish(ddof=0, group=False)
        torch.cuda.synchronize()
        for name in ("actions", "logprobs", "rewards", "masks", "obs", "values") + (("valid",) if inband else ()):
            got = torch.cat([getattr(r, name) for r in sh.ros], dim=-1)
            assert torch.equal(got, getattr(one, name)), (rep, name)
        assert torch.equal(torch.cat([o[0] for o in outs], dim=-1), ret1)
        assert torch.allclose(torch.cat([o[1] for o in outs], dim=-1), adv1, rtol=0, atol=1e-10)
        assert torch.equal(sh.last_returns(), one.env.last_return)
    assert float(one.masks.min()) == 0.0  # episodes did end and reset inside the horizons


def test_sharded_rollout_keeps_a_handles_fused_switch(aq):
    """StreamShardedRollout only asks for 64-env blocks (qenv_set_ppo_fused_block); a shard env that was switched to the
    separate launches for an A/B run stays switched (the round-3 code forced the fused launch back on)."""
    import torch
    w = aq.ActorCriticWeights.random(128, 75, 4, device="cuda", seed=2)
    made = []

    def make_env(lo, hi):
        env = aq.BatchedQuad(hi - lo, 0.01, 40, training=True, direct_control=1, T=5)
        env.set_ppo_fused(len(made) % 2 == 0)   # shard 0 fused, shard 1 the separate launches
        made.append(env)
        return env
    sh = aq.StreamShardedRollout(make_env, 256, w, 8, shards=2, seed=1, store_obs=True, auto_reset=True, inband=True)
    assert [r.env.ppo_fused for r in sh.ros] == [(True, 64), (False, 64)]
    env = aq.BatchedQuad(64, 0.01, 40, training=True, direct_control=1, T=5)
    assert env.ppo_fused == (True, None)
    env.set_ppo_fused(True, block=32); assert env.ppo_fused == (True, 32)
    env.set_ppo_fused_block(None); assert env.ppo_fused == (True, None)
    env.set_ppo_fused(False); env.set_ppo_fused_block(32); assert env.ppo_fused == (False, 32)
    env.set_ppo_fused_block(64); env.set_ppo_fused(True); assert env.ppo_fused == (True, 64)   # a plain on / off keeps the block
    env.set_ppo_fused(False); assert env.ppo_fused == (False, 64)
    with pytest.raises(ValueError):
        env.set_ppo_fused_block(48)


def test_ppo_rollout_validates_up_front_on_every_path(aq):
    """qenv_ppo_rollout checks its argument block ONCE before anything is launched, with the list qenv_ppo_step uses --
    whether the horizon then runs as the fused launch or as per-step calls (hidden 64): a non-positive std, a null weight
    pointer, a slice layout without the in-band schedule and a window of the wrong length all fail with QENV_ERR_ARG and
    leave the env untouched."""
    import torch
    from autonomous_quadrotor_environment_amd import _lib as L
    for hidden in (128, 64):
        w = aq.ActorCriticWeights.random(hidden, 75, 4, device="cuda", seed=3)
        env = aq.BatchedQuad(128, 0.01, 40, training=True, direct_control=1, T=5)
        env.reset(env.sample_init_states(3, 0, 0), return_history=False, push_window=True)
        ro = aq.PolicyRollout(env, w, 4, store_obs=True, auto_reset=True, seed=1, inband=True)
        state0 = env.state.clone()
        W = w.head("actor")

        def args(**over):
            kw = dict(k_in=75, hidden=hidden, W=[x.data_ptr() for x in W], std=0.1, inband=1, obs_rows_from=0, obs_slices=0)
            kw.update(over)
            return L.PpoStepArgs(kw["k_in"], kw["hidden"], *kw["W"], kw["std"], None, 1, 0, 0, None, ro.actions[0].data_ptr(),
                                 ro.logprobs[0].data_ptr(), None, ro.rewards[0].data_ptr(), ro.done.data_ptr(),
                                 ro.masks[0].data_ptr(), ro.valid[0].data_ptr(), kw["inband"], 0, kw["obs_rows_from"], kw["obs_slices"])
        import ctypes as C
        env._bind()
        bad = [args(std=0.0), args(W=[W[0].data_ptr(), None] + [x.data_ptr() for x in W[2:]]), args(inband=0, obs_slices=1),
               args(k_in=60), args(obs_rows_from=75)]
        for a in bad:
            assert L.lib.qenv_ppo_rollout(env._h, C.byref(a), 4) == -1, hidden   # QENV_ERR_ARG
        torch.cuda.synchronize()
        assert torch.equal(env.state, state0)
        assert L.lib.qenv_ppo_rollout(env._h, C.byref(args()), 4) == 0
        torch.cuda.synchronize()
        assert not torch.equal(env.state, state0)


def test_rccl_itself_runs_the_paths_collectives_on_one_rank():
    """The N > 1 path's collectives through RCCL (backend "nccl") on the one GPU of this box: a one-rank communicator loads
    librccl, is created on the device and runs the all-gather of the episode returns, the MAX / SUM all-reduces and the barrier as
    device operations (tests/tools/rccl_one_rank.py, its own process: the process group is global state).  The multi-rank logic is
    covered under gloo (tests/test_distributed_gloo.py, the --gpus 2 bench tests); this shows the RCCL half of the path is usable
    on the image and the hardware."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["backend"] == "nccl" and d["world"] == 1 and d["all_gather_512KB_us"] > 0
