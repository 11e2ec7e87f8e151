"""GPU: a short run of every differential fuzzer under tests/tools (HIP path vs the oracle -- the learner: vs torch autograd
in fp64 -- over random points of the configuration space; the long runs are recorded under profiles/r01_fuzz_*.jsonl and profiles/r02_parity/).  The scripts are executed in this
process (runpy), with a seed the recorded runs did not use."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")


@pytest.mark.parametrize("script,args,handle", [(s_, a_, None) for s_, a_ in [
    ("fuzz_parity.py", ["101", "100"]),             # fp64 step/reset across variant x integrator x mode x T x ...
    ("fuzz_parity_f32.py", ["101", "60", "0.01"]), # fp32 one-step error at the reference's step size
    ("fuzz_controllers.py", ["101", "60"]),        # fused LQR / PID / position-PD / constant rollouts
    ("fuzz_robust.py", ["101", "30"]),              # robust_control: perturbed plants + gust tables
    ("fuzz_policy_rollout.py", ["101", "20"]),      # PPO worker loop with masked auto-reset
    ("fuzz_outer_loops.py", ["101", "30"]),         # vel_pid / pos_pi inside the window push
    ("fuzz_sensor.py", ["101", "30"]),              # sensor model
    ("fuzz_visual_reward.py", ["101", "30"]),       # landing reward around its decision boundaries
    ("fuzz_quad_dropin.py", ["101", "25"]),         # the N = 1 numpy-in / numpy-out class over random API sequences
    ("fuzz_mlp.py", ["101", "50"]),                 # every MLP forward path vs torch fp64 (ring phases, T, hidden, K, N)
    ("fuzz_gae.py", ["101", "60"]),                 # GAE + normalisation, atomic and ordered moments
    ("fuzz_learner.py", ["101", "30"]),             # learner gradients vs torch autograd (fp64), random batch shapes
    ("fuzz_fused_rollout.py", ["101", "16"]),       # the fused rollout launch vs the per-step launches, bit for bit
]] + [
    # the parity-qualified lean handle (fp64 rows, lean fp64 arithmetic) through the same fp64 fuzzers and tolerances
    ("fuzz_parity.py", ["102", "100"], "lean"),
    ("fuzz_controllers.py", ["102", "40"], "lean"),
    ("fuzz_policy_rollout.py", ["102", "20"], "lean"),
    ("fuzz_outer_loops.py", ["102", "20"], "lean"),
])
def test_fuzzer_short_run(script, args, handle, capsys, monkeypatch):
    argv = sys.argv
    sys.argv = [script] + args
    if handle:
        monkeypatch.setenv("QENV_FUZZ_HANDLE", handle)
    else:
        monkeypatch.delenv("QENV_FUZZ_HANDLE", raising=False)
    try:
        with pytest.raises(SystemExit) as ex:
            runpy.run_path(os.path.join(TOOLS, script), run_name="__main__")
    finally:
        sys.argv = argv
    out = capsys.readouterr().out.strip().splitlines()
    assert ex.value.code in (0, None), "\n".join(out[-5:])
