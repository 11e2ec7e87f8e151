"""GPU box: actor forward (75 -> H -> H -> 4) launch time by batch size.  Run once with QENV_MLP_SMALL_TILES=0 (always
the LDS-staged k_mlp) and once with QENV_MLP_SMALL_TILES=1000000 (always the split k_mlp_small) to place the switch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = aq.ActorCriticWeights.random(hidden, 75, 4, device="cuda", seed=1)
res = {}
for N in (1, 32, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
    env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5, dtype=torch.float32)
    env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
    out = torch.empty(4, N, device="cuda")
    for _ in range(20):
        env.mlp_forward(w, "actor", out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        env.mlp_forward(w, "actor", out=out)
    b.record(); torch.cuda.synchronize()
    res[N] = round(a.elapsed_time(b) * 1000 / 200, 2)
print(json.dumps(dict(hidden=hidden, small_tiles=os.environ.get("QENV_MLP_SMALL_TILES"), us=res)))
