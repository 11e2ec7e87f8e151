"""one-launch gradient kernel vs the round-4 path and torch fp64 autograd"""
import os, sys, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.nn as nn
import autonomous_quadrotor_environment_amd as aq

def net(ws, final_tanh):
    W1, b1, W2, b2, W3, b3 = ws
    m = nn.Sequential(nn.Linear(W1.shape[1], W1.shape[0]), nn.Tanh(), nn.Linear(W2.shape[1], W2.shape[0]), nn.Tanh(),
                      nn.Linear(W3.shape[1], W3.shape[0]), *([nn.Tanh()] if final_tanh else [])).to("cuda", torch.float64)
    with torch.no_grad():
        for lin, (W, b) in zip([m[0], m[2], m[4]], ((W1, b1), (W2, b2), (W3, b3))):
            lin.weight.copy_(W.double()); lin.bias.copy_(b.double())
    return m

for (n, N, det) in ((1, 32, False), (3, 200, False), (2, 33, True), (8, 5000, False), (4, 4097, True)):
    torch.manual_seed(n * 1000 + N + int(os.environ.get('GRAD_CHECK_SEED', '0')))
    w = aq.ActorCriticWeights.random(128, 75, 4, seed=3); w.std = 0.1
    obs = torch.randn(n, 75, N, device="cuda")
    act_net, cri_net = net(w.actor, True), net(w.critic, False)
    x = obs.permute(0, 2, 1).double()
    with torch.no_grad():
        mean0 = act_net(x)
    actions = (mean0 + 0.1 * torch.randn_like(mean0)).permute(0, 2, 1).float().contiguous()
    old_lp = torch.distributions.Normal(mean0 + 0.03 * torch.randn_like(mean0), 0.1).log_prob(actions.permute(0, 2, 1).double())
    old_lp = old_lp.permute(0, 2, 1).float().contiguous()
    adv = torch.randn(n, N, device="cuda"); rets = 2 * torch.randn(n, N, device="cuda")
    valid = (torch.rand(n, N, device="cuda") > 0.2).float() if det else None
    res = {}
    for one in (True, False):
        ln = aq.PpoLearner(w, K_epochs=1, deterministic=det)
        ln.one_launch = one
        loss = ln.gradients(obs, actions, old_lp, adv, rets, valid=valid)
        torch.cuda.synchronize()
        res[one] = ([ln._state[(h, k)][0].double().clone() for h in ("actor", "critic") for k in range(6)], {k: float(v) for k, v in loss.items()})
    mean = act_net(x); value = cri_net(x).squeeze(-1)
    dist = torch.distributions.Normal(mean, 0.1)
    lp = dist.log_prob(actions.permute(0, 2, 1).double()).sum(-1)
    ratio = torch.exp(lp - old_lp.permute(0, 2, 1).double().sum(-1))
    a_ = adv.double()
    wv = torch.ones_like(a_) if valid is None else valid.double()
    B = wv.sum()
    total = ((-torch.min(ratio * a_, torch.clamp(ratio, 0.8, 1.2) * a_)) * wv).sum() / B + 0.5 * (((value - rets.double()) ** 2) * wv).sum() / B
    total.backward()
    refs = [p.grad for netw in (act_net, cri_net) for l in (netw[0], netw[2], netw[4]) for p in (l.weight, l.bias)]
    names = [f"{h}.{k}" for h in ("actor", "critic") for k in ("W1", "b1", "W2", "b2", "W3", "b3")]
    out = {}
    for i, nm in enumerate(names):
        r = refs[i]; sc = float(r.abs().max()) + 1e-30
        out[nm] = (float((res[True][0][i] - r).abs().max()) / sc, float((res[False][0][i] - r).abs().max()) / sc)
    print(json.dumps(dict(n=n, N=N, det=det, loss_new=res[True][1]["actor"] + res[True][1]["critic"], loss_old=res[False][1]["actor"] + res[False][1]["critic"],
                          loss_ref=float(total.detach()), err_new_old={k: [float(f"{a:.2e}"), float(f"{b:.2e}")] for k, (a, b) in out.items()})), flush=True)
    if det:
        ln = aq.PpoLearner(w, K_epochs=1, deterministic=True)
        ln.gradients(obs, actions, old_lp, adv, rets, valid=valid)
        g2 = [ln._state[(h, k)][0].double().clone() for h in ("actor", "critic") for k in range(6)]
        print("deterministic repeat bit-equal:", all(torch.equal(a, b) for a, b in zip(g2, res[True][0])))
