"""CPU (no GPU): the C-ABI library loads and exports every symbol include/qenv.h declares,
argument validation works without touching a device, and the host-side logic (RNG replay,
shard ranges, seeded shard-invariant initial states) is right."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from .conftest import ROOT, golden, rel_err


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import autonomous_quadrotor_environment_amd as aq
    return aq


def test_library_exports_every_declared_symbol(built):
    from autonomous_quadrotor_environment_amd import _lib
    header = open(os.path.join(ROOT, "include", "qenv.h")).read()
    declared = set(re.findall(r"\b(qenv_[a-z_]+)\s*\(", header)) - {"qenv_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.qenv_abi_version() == _lib.QENV_ABI_VERSION


def test_argument_validation_without_a_device(built):
    from autonomous_quadrotor_environment_amd import _lib as L
    cfg = L.QenvConfig(1024, 0.01, 1000, 5, 1, 1, 1, L.MAIN, L.RK45, L.F32, 0)
    n32 = L.lib.qenv_workspace_bytes(C.byref(cfg))
    cfg.dtype = L.F64
    n64 = L.lib.qenv_workspace_bytes(C.byref(cfg))
    # 88 rows of T (61 + 12 robust parameters + 6 gust + 9 outer loop) + 6 int rows + the 256-byte tail of the two
    # reset-list counters (their own slot, valid for n_envs = 1 too) + two byte-per-env flag arrays (256-byte aligned)
    # + 15*T float rows, 256-byte aligned sections
    assert n32 == 88 * 1024 * 4 + 6 * 1024 * 4 + 256 + 2 * 1024 + 75 * 1024 * 4
    assert n64 == 88 * 1024 * 8 + 6 * 1024 * 4 + 256 + 2 * 1024 + 75 * 1024 * 4
    one = L.QenvConfig(1, 0.01, 1000, 5, 1, 1, 1, L.MAIN, L.RK45, L.F32, 0)
    assert L.lib.qenv_workspace_bytes(C.byref(one)) == 512 + 256 + 256 + 512 + 512  # rows, int rows, counters, flags, ring
    lean = L.QenvConfig(1024, 0.01, 1000, 5, 1, 1, 1, L.MAIN, L.RK45, L.F64, 2)
    assert L.lib.qenv_workspace_bytes(C.byref(lean)) == n64
    assert L.lib.qenv_create(C.byref(L.QenvConfig(4, 0.01, 10, 1, 1, 1, 1, L.MAIN, L.RK45, L.F32, 2)), C.c_void_p(256), 1 << 20, None,
                             C.byref(C.c_void_p())) == -1 and b"LEAN" in L.lib.qenv_last_error()
    bad = L.QenvConfig(0, 0.01, 1000, 1, 1, 1, 1, 0, 0, 0, 0)
    assert L.lib.qenv_workspace_bytes(C.byref(bad)) == 0
    h = C.c_void_p()
    assert L.lib.qenv_create(C.byref(cfg), None, 0, None, C.byref(h)) == -1
    assert b"null" in L.lib.qenv_last_error()
    assert L.lib.qenv_step(None, None, None, None, None, None, 0) == -1
    assert L.lib.qenv_gae(0, 0, 10, None, None, None, None, None, None, None) == -1
    # the entry points of the section-8(f) rows refuse a null handle / null operands the same way, before any HIP call
    assert L.lib.qenv_set_robust(None, 1) == -1 and b"qenv_set_robust" in L.lib.qenv_last_error()
    assert L.lib.qenv_robust_params(None, None, 0, 0, None) == -1
    assert L.lib.qenv_robust_gusts(None, None, 1, 1, 0, 0) == -1
    assert L.lib.qenv_outer_velpid(None, None, 0) == -1 and L.lib.qenv_outer_pospi(None, None, 0.0, 0.0, 0.0) == -1
    assert L.lib.qenv_outer_reset(None, None) == -1
    assert L.lib.qenv_sensor_read(None, None, None, 0, 0, 0, 1, None, None) == -1
    assert L.lib.qenv_sensor_ops(None, None, None, 1, None, 1, None, None) == -1
    assert L.lib.qenv_ppo_step(None, None) == -1 and L.lib.qenv_ppo_flush(None) == -1
    assert L.lib.qenv_attach_sensor(None, None, None, 0, 0, 0, None, None, 0) == -1
    assert L.lib.qenv_dense_tail_chunks(64, 0, 75, 128, 4, 1, *([None] * 8), None) == -1
    assert L.lib.qenv_mlp_backward(0, 1, 75, 128, 4, 1, *([None] * 13), None) == -1
    with pytest.raises(L.QenvError):
        L.check(L.lib.qenv_destroy(None), "qenv_destroy")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "autonomous_quadrotor_environment_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "__init__.py" and "oracle" not in src, (f,)


def test_only_the_allowed_places_load_the_oracle():
    """oracle/ is test infrastructure: besides tests/ (incl. tests/tools), only __graft_entry__.smoke() and bench.py's
    cpu_baseline leg may load it -- tools/, examples/ and the package never do."""
    for sub in ("tools", "examples", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".c", ".h", ".hip")):
                    assert "quad_oracle" not in open(os.path.join(dirpath, f)).read(), (sub, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("quad_oracle") == 1 and bench.index("quad_oracle") > bench.index("def cpu_baseline")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert entry.index("def smoke") < entry.rindex("quad_oracle") or "oracle" in entry[:entry.index("def smoke")]  # build() compiles it


def test_random_init_states_replay_reference_draw_order(built):
    g = golden("g5_reset")
    for seed in range(16):
        x0 = built.random_init_states(seed, 1)[0]
        assert rel_err(x0, g["x0_d1_T1"][seed]) < 1e-14
    two = built.random_init_states(np.random.RandomState(7), 2)
    assert rel_err(two[0], g["x0_d1_T1"][7]) < 1e-14 and not np.allclose(two[0], two[1])


def test_fast_init_is_shard_invariant(built):
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    from autonomous_quadrotor_environment_amd.distributed import shard_range
    n = 10000
    full = fast_random_init_states(5, n)
    assert full.shape == (n, 13)
    assert np.allclose(np.linalg.norm(full[:, 6:10], axis=1), 1.0)
    assert np.all(np.abs(full[:, 0:5:2]) <= 2.5) and np.all(np.abs(full[:, 1:6:2]) <= 5)
    assert np.all(full[:, 10:13] <= 7.5) and np.all(full[:, 10:13] >= -15)
    for world in (2, 3, 8):
        parts = [fast_random_init_states(5, n, *shard_range(n, r, world)) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), full)
    assert not np.array_equal(fast_random_init_states(6, n), full)


def test_shard_range_partitions():
    from autonomous_quadrotor_environment_amd.distributed import shard_range
    for n, w in ((1048576, 8), (10, 3), (7, 8), (524288, 4)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_gpu_only_classes_refuse_cpu(built):
    import torch
    with pytest.raises(RuntimeError):
        built.BatchedQuad(4, 0.01, 100, device="cpu")


def test_mission_trajectories_match_reference(built):
    """G16: mission_control.mission (step, velocity ramp, sine, spiral incl. the end-of-trajectory drift)."""
    import numpy as _np
    g = golden("g16_mission")
    M = built.mission
    cases = {"step": (lambda q: q.gen_trajectory(700, 300, _np.array([1.0, -0.5, 0.8])), 750),
             "ramp_vel": (lambda q: q.gen_trajectory(200, 200, _np.zeros(3), velocity=_np.array([0.5, 0.2, -0.1])), 300),
             "sin": (lambda q: q.sin_trajectory(600, 1.5, 0.1, _np.array([0.5, 0, 0]), _np.array([1.0, 0.5, 0])), 650),
             "spiral": (lambda q: q.spiral_trajectory(300, 600, 0.3, 1.2, 0.8, _np.array([0, 0, 0.0])), 620)}
    for name, (f, n) in cases.items():
        q = M(0.01)
        f(q)
        got = _np.array([q.get_error(0) for _ in range(n)])
        assert _np.allclose(got, g[name], rtol=0, atol=1e-12), name
        q2 = M(0.01)
        f(q2)
        tab = q2.table(n)
        assert _np.allclose(tab[:, 0:3], g[name][:, 0:5:2], atol=1e-12) and _np.allclose(tab[:, 3:6], g[name][:, 1:6:2], atol=1e-12)


def test_header_is_plain_c_and_a_c_host_links(built, tmp_path):
    """include/qenv.h compiles as C99 and as C++11 (-Wall -Wextra -pedantic clean) and a C host links libqenv.so."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text('#include "qenv.h"\n#include <stdio.h>\nint main(void){ qenv_config c; c.n_envs = 0; '
                   'printf("%zu %d\\n", qenv_workspace_bytes(&c), qenv_abi_version()); '
                   'return qenv_abi_version() == QENV_ABI_VERSION ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "autonomous_quadrotor_environment_amd")
    for cc, std, extra in (("gcc", "-std=c99", []), ("g++", "-std=c++11", ["-x", "c++"])):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, *extra, "-c", str(src), "-o",
                            str(tmp_path / "host.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = tmp_path / "host"
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-L", libdir, "-lqenv", f"-Wl,-rpath,{libdir}",
                        "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[0] == "0", (r.stdout, r.stderr)  # n_envs = 0 is rejected: 0 bytes


def test_bench_refuses_to_report_one_rank_for_an_n_gpu_request():
    """`bench.py --gpus N` must either run N ranks or fail loudly: with no GPU visible the self-launching parent exits
    non-zero before any rank starts, a WORLD_SIZE that contradicts --gpus does too, and neither prints a result line."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QENV_DIST_BACKEND")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "20", "--warmup", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout and "--gpus 2" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "20", "--warmup", "5"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout and "WORLD_SIZE=2" in r.stderr


def test_lean_fastmath_accuracy(tmp_path):
    """The lean fp64 handle's elementary functions (csrc/qenv_fastmath.hpp: hardware seed + Newton, short minimax
    polynomials) on the host with seeds of the hardware's accuracy: within a few ulp of libm over 2 M random arguments
    incl. the branch points, with libm's edge cases (atan2 at the origin / on the axes, asin at and beyond +-1)."""
    import json
    import subprocess
    exe = str(tmp_path / "fm_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "fastmath_host_check.cpp")])
    r = json.loads(subprocess.check_output([exe], text=True))
    assert r["edges"] is True
    assert r["atan2_abs"] < 1e-15 and r["asin_abs"] < 1e-15 and r["sin_abs"] < 5e-16 and r["cos_abs"] < 5e-16
    assert r["tan_rel"] < 1e-15 and r["div_rel"] < 3e-16 and r["sqrt_rel"] < 3e-16 and r["rsqrt_rel"] < 5e-16


def test_weight_versions_belong_to_the_storage_not_the_wrapper():
    """ActorCriticWeights.version_of keys an env's prepared weight image: every wrapper of the same tensors shares one
    cell (an alias held by an evaluation env sees the learner's touch()), a wrapper that copies gets its own, and the
    cells follow the tensors into the learner's flat parameter buffer.  Host logic only (CPU tensors, no launch)."""
    import torch
    from autonomous_quadrotor_environment_amd.ppo import ActorCriticWeights, PpoLearner
    rng = np.random.RandomState(0)
    mk = lambda o: [rng.randn(16, 75), rng.randn(16), rng.randn(16, 16), rng.randn(16), rng.randn(o, 16), rng.randn(o)]
    w = ActorCriticWeights(mk(4), mk(1), device="cpu")
    alias = ActorCriticWeights(w.actor, None, device="cpu")
    assert alias.actor[0].data_ptr() == w.actor[0].data_ptr()
    v0 = w.version_of("actor")
    assert alias.version_of("actor") == v0 and w.version_of("critic") != v0
    w.touch()
    assert alias.version_of("actor") == w.version_of("actor") != v0
    alias.touch()   # either wrapper may announce a change
    assert alias.version_of("actor") == w.version_of("actor")
    copy = ActorCriticWeights([t.double() for t in w.actor], None, device="cpu")
    assert copy.actor[0].data_ptr() != w.actor[0].data_ptr() and copy.version_of("actor") != w.version_of("actor")
    before = w.version_of("actor")
    learner = PpoLearner(w, K_epochs=1)   # re-homes the tensors into one flat buffer
    assert w.actor[0].data_ptr() == learner.flat_p.data_ptr() and w.version_of("actor") != before
    late = w.actor_only()
    assert late.version_of("actor") == w.version_of("actor")
    w.touch()
    assert late.version_of("actor") == w.version_of("actor")
    # the alias made BEFORE the learner moved the tensors holds the old copies: it says so instead of serving stale weights
    with pytest.raises(RuntimeError, match="not the trained ones"):
        alias.version_of("actor")
    # the registry holds a cell under its home address only and weakly: the address the tensors left is free for an unrelated
    # tensor, and the entries of collected wrappers are gone
    from autonomous_quadrotor_environment_amd import ppo as P
    old_home = alias.actor[0].data_ptr()
    assert old_home not in P._VERSION_CELLS
    stranger = ActorCriticWeights(alias.actor, None, device="cpu")   # same address as the stale copies, not the learner's cell
    assert stranger.version_of("actor") != w.version_of("actor")
    n_before = len(P._VERSION_CELLS)
    del stranger, copy
    import gc
    gc.collect()
    assert len(P._VERSION_CELLS) < n_before


def test_probe_bound_is_conservative(tmp_path, oracle):
    """csrc/qenv_probe_bound.hpp proves, from y, f0 and h0 alone, that scipy's select_initial_step would get h1 >= t_bound, and
    the kernels then skip the evaluation f(y + h0 f0) that only feeds d2.  The proof is a Lipschitz bound on d2: here the header
    is built for the host and its bound is compared with the ACTUAL d2 from the oracle's drone_eq over wide random states
    (hover-like to far outside the done box, both control modes, both variants' drag moments): bound >= d2 always, and the
    proof succeeds on most of them (so the skip is what usually runs)."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "probe_bound.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "tools", "probe_bound_host.cpp")])
    pb = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    pb.pb_sum.restype = C.c_double
    pb.pb_sum.argtypes = [dp, dp, dp, dp, C.c_double]
    pb.pb_skippable.restype = C.c_int
    pb.pb_skippable.argtypes = [dp, dp, dp, dp, C.c_double, C.c_double, C.c_double]
    pb.pb_skippable_f.restype = C.c_int
    pb.pb_skippable_f.argtypes = pb.pb_skippable.argtypes
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    M, G, KF, IR, D = 1.03, 9.82, 1.435e-5, 5e-5, 0.26
    J = np.array([16.83e-3, 16.83e-3, 28.34e-3])
    RHO, CD, BEAM = 1.2041, 1.1, 0.05
    KDZ = 0.5 * RHO * CD * 4 * BEAM * D
    CM0 = RHO * CD * BEAM * D / 10
    rng = np.random.RandomState(7)
    tb, thr = 0.01, 0.01 / 0.01 ** 5
    rms = lambda x: float(np.sqrt(np.mean(x * x)))
    worst, n_proven, n_proven_f, n = np.inf, 0, 0, 0
    for variant, sxx in (("main", D ** 3 * 2025 / 729), ("opt", D * D * 285 / 81)):
        for direct in (1, 0):
            env = oracle.OracleQuad(tb, 1000, training=False, direct_control=direct, T=1, variant=variant)
            for k in range(700):
                scale = 10.0 ** rng.uniform(-3, 0.3)          # hover-like ... twice the done box
                y = np.zeros(13)
                y[0:5:2] = rng.normal(0, 2, 3) * (rng.rand() < 0.8)
                y[1:6:2] = rng.normal(0, 5, 3) * scale
                q = rng.normal(0, 1, 4) if k % 3 else np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.3 * scale, 4)
                y[6:10] = q / np.linalg.norm(q) * (1 + rng.normal(0, 1e-3))
                y[10:13] = rng.normal(0, 10, 3) * scale
                if direct:
                    a = np.clip(rng.normal(0, 0.6, 4), -1, 1) if k % 4 else np.zeros(4)
                    f = (a + 1) * 2 * M * G / 8
                    w = np.sqrt(f / KF)
                    F = f.sum()
                else:
                    F = M * G * (1 + rng.normal(0, 0.3)); a = np.array([F, *rng.normal(0, 0.05, 3)])
                    w = np.sqrt(np.clip(F / 4 / KF + rng.normal(0, 1e3, 4), 0, None))
                    env.set_w(w)
                omr = ((w[1] - w[0]) + (w[3] - w[2])) * IR
                f0 = env.drone_eq(0.0, y, a)
                sc = 1e-6 + np.abs(y) * 1e-3
                d0, d1 = rms(y / sc), rms(f0 / sc)
                h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
                h0 = min(h0, tb)                                     # select_initial_step's clamp
                if k % 5 == 0:
                    h0 = min(h0 * 10.0 ** rng.uniform(-2, 2), tb)    # the bound holds for ANY 0 < h0 <= h_max = t_bound
                f1 = env.drone_eq(h0, y + h0 * f0, a)
                d2 = rms((f1 - f0) / sc) / h0
                # (kd: the largest of the three drag coefficients -- KDZ for this plant; rk45_initial_step takes the maximum)
                plant = np.array([abs(F), abs(omr) / J[0], CM0 * sxx / J[0], 2 * CM0 * sxx / J[2], KDZ, 1 / M,
                                  abs((J[2] - J[1]) / J[0]), abs((J[0] - J[2]) / J[1]), abs((J[1] - J[0]) / J[2])])
                s = pb.pb_sum(P(y), P(f0), P(1 / sc), P(plant), tb)
                d2b = np.sqrt(s / 13)
                assert d2b >= d2 * (1 - 1e-9) - 1e-6, (variant, direct, k, d2b, d2)
                worst = min(worst, d2b / max(d2, 1e-300))
                ok = pb.pb_skippable(P(y), P(f0), P(1 / sc), P(plant), tb, d1, thr)
                if ok:   # what the kernels rely on: proven => scipy's h1 >= t_bound
                    assert max(d1, d2) <= thr and d1 > 1e-15
                n_proven += ok; n += 1
                # the float instantiation (f32 / mixed handles): its rounding sits far inside the factor 2 the comparison keeps
                okf = pb.pb_skippable_f(P(y), P(f0), P(1 / sc), P(plant), tb, d1, thr)
                if okf:
                    assert max(d1, d2) <= thr and d1 > 1e-15
                n_proven_f += okf
    assert worst >= 1.0 and n_proven > 0.5 * n and n_proven_f > 0.5 * n, (worst, n_proven, n_proven_f, n)
    # the preconditions: a near-zero d1 (scipy's other branch) and a degenerate quaternion are never "proven"
    y = np.zeros(13); y[6] = 1.0
    z = np.zeros(13)
    plant = np.array([M * G, 0, 1e-3, 2e-3, KDZ, 1 / M, 0.68, 0.68, 0.0])
    assert pb.pb_skippable(P(y), P(z), P(np.full(13, 1e3)), P(plant), tb, 0.0, thr) == 0
    y[6] = 0.3
    assert pb.pb_skippable(P(y), P(np.full(13, 0.1)), P(np.full(13, 1e3)), P(plant), tb, 50.0, thr) == 0


def test_grad5_lds_address_maps():
    """k_mlp_grad_v5's LDS images (csrc/qenv_grad5.hpp): tools/lds_layout_grad5.py replays the kernel's lane address formulas on a
    model of the LDS and of ds_read_b64_tr_b16 -- every fragment is the MFMA operand the kernel claims, and only the accumulator
    fragments' 8-byte stores conflict (two-way) under MI355X_MICROARCH.md's bank rules."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lds_layout_grad5.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wrong elements in total 0" in r.stdout and r.stdout.strip().endswith("OK")
