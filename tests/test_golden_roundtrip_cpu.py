"""The committed fixtures and their generators round-trip (twelve fixtures had fallen behind the generator's
`meta`).  Build container only -- the generators import the reference from /root/reference; skipped where it is absent (GPU box).
One small fixture per generator is regenerated into a temporary directory (MNRF_GOLDEN_OUT) and compared with the committed file:
every array bit for bit AND the JSON meta."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# (generator module, statement run after importing it as G, fixtures it writes)
CASES = [
    ("make_golden", "G.g1(); G.eval_case('g7_eval_l1', 96, G.W.STRADDLE, 1)", ["g1_embedding", "g7_eval_l1"]),
    ("make_golden", "G.render_case('g3_coarse64_test', 256, 3, 0, test_time=True, compute_normal=False, keep_per_sample=False)",
     ["g3_coarse64_test"]),
    ("make_golden_config1", "G.eval_case('g15_c1_eval_l1', 96, G.STRADDLE_C1)", ["g15_c1_eval_l1"]),
    ("make_golden_variants", "G.render('g13_plain_nerf_test', False, False, True, False)", ["g13_plain_nerf_test"]),
    ("make_golden_nemb", "G.case('g16_nemb_6_2_train_grads', 64)", ["g16_nemb_6_2_train_grads"]),
    ("make_golden_rays", "G.main()", ["g12_rays_37x53", "g12_rays_64x64"]),
    ("make_golden_flags", "sys.argv[1:] = ['g9b_detach_ref_color']; G.main()", ["g9b_detach_ref_color"]),
    ("make_golden_tcnn", "G.grid_offsets()", ["g17_grid_offsets"]),
    ("make_golden_totalloss", "G.run_case('g10_loss_default', 1, dict(use_plane_consistent_loss=True), False, 5)", ["g10_loss_default"]),
    ("make_golden_truth64", "G.truth('g3_coarse64_train')", ["g14_truth64_g3_coarse64_train"]),
    ("make_golden_trained_capture", "sys.argv[1:] = ['g11_trained_psnr']; G.main()", ["g11_trained_psnr"]),
    ("make_golden_spike", "G.main()", ["g18_grad_spike"]),
]


def _same(a, b, path=""):
    """JSON values equal, floats to the last bit (json round-trips a Python float exactly)."""
    assert type(a) is type(b) or {type(a), type(b)} <= {int, float}, (path, a, b)
    if isinstance(a, dict):
        assert sorted(a) == sorted(b), (path, sorted(set(a) ^ set(b)))
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference tree is not on this machine (GPU box)")
@pytest.mark.parametrize("module,stmt,names", CASES, ids=[f"{c[0]}:{c[2][0]}" for c in CASES])
def test_generator_reproduces_committed_fixture(tmp_path, module, stmt, names):
    code = f"import sys; sys.path.insert(0, {GOLDEN!r}); import {module} as G; {stmt}"
    env = dict(os.environ, MNRF_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=GOLDEN, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for name in names:
        new, old = np.load(tmp_path / f"{name}.npz"), np.load(os.path.join(GOLDEN, f"{name}.npz"))
        assert sorted(new.files) == sorted(old.files), (name, sorted(set(new.files) ^ set(old.files)))
        for k in old.files:
            if k == "meta":
                _same(json.loads(str(new[k])), json.loads(str(old[k])), name + ".meta")
            else:
                assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape, (name, k)
                assert np.array_equal(new[k], old[k], equal_nan=True), (name, k)
