"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mnrf.h declares,
and validates arguments before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mnrf.h")


@pytest.fixture(scope="module")
def L():
    from mirror_nerf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_exports_match_header(L):
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(mnrf_[a-z0-9_]+)\s*\(", text))
    from mirror_nerf_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), f"libmnrf_hip.so lacks {name}"


def test_packed_size_matches_layout(L):
    # fp32 streams: forward tiles 2640 + bias block 3072 floats + backward tiles 1920 + head-backward tiles 704;
    # split-f16 streams: 1312 forward + 960 trunk-backward + 352 head-backward hi/lo tile pairs of 2 KiB (mnrf_layout.h)
    assert L.mnrf_packed_floats() == 2640 * 256 + 3072 + 1920 * 256 + 704 * 256 + (1312 + 960 + 352 + 1328) * 512 + 16384   # + tail pad
    assert L.mnrf_version() >= 1


def test_argument_validation_without_gpu(L):
    null = None
    assert L.mnrf_field_forward(null, 0, 128, null, 3, null, null, 1, null, 27, null, null, null, null, null,
                                null, null) < 0
    assert b"packed" in L.mnrf_last_error()
    assert L.mnrf_sample_fine(null, null, 4, 2, null, 0, 16, null, null) < 0      # S < 3
    assert L.mnrf_sample_fine(null, null, 4, 300, null, 0, 16, null, null) < 0    # S > 256
    assert L.mnrf_composite(null, 4, 64, null, null, null, null, null, null, null, 0, null, null, null, null,
                            null, null, null, null, null, null) < 0
    assert L.mnrf_embed(null, -1, 3, 4, null, null) < 0
    # loss reductions / metric: null argument block, missing outputs, bad sizes
    assert L.mnrf_total_loss(null, null, null) < 0
    assert b"mnrf_total_loss" in L.mnrf_last_error()
    from mirror_nerf_amd.losses import _Args
    import ctypes
    a = _Args()
    a.n_rays = 0
    ws = (ctypes.c_float * 64)()
    assert L.mnrf_total_loss(ctypes.byref(a), ctypes.cast(ws, ctypes.c_void_p), null) < 0      # n_rays must be positive
    a.n_rays = 4
    assert L.mnrf_total_loss(ctypes.byref(a), ctypes.cast(ws, ctypes.c_void_p), null) < 0      # out / targets / rays missing
    assert b"required" in L.mnrf_last_error()
    assert L.mnrf_mse_psnr(null, null, null, 10, 1, null, null, null) < 0
    assert L.mnrf_loss_workspace_floats(1024, 64, 192, 0) > 1024
    assert L.mnrf_mse_blocks() >= 1
    # round-3 entry points: the same contract (null pointers / sizes are refused with a message; sizes are plain arithmetic)
    import ctypes as C
    assert L.mnrf_field_composite_fused(null, 4, null, null, null, 27, 0, null, null, null, null, null, null, null, null) < 0
    assert b"mnrf_field_composite_fused" in L.mnrf_last_error()
    assert L.mnrf_fused_samples_per_ray() == 192
    assert L.mnrf_dw_planes(0, null, null, null, null, null, null, 0, null) < 0          # 1..8 evaluations
    assert L.mnrf_dw_planes(1, null, null, null, null, null, null, 0, null) < 0
    assert b"mnrf_dw_planes" in L.mnrf_last_error()
    assert L.mnrf_train_planes_bytes(128) == 4 * 174 * 2048 and L.mnrf_train_planes_bytes(129) == 8 * 174 * 2048
    assert L.mnrf_train_dy_planes_bytes(128) == 4 * 172 * 2048
    one = (C.c_int64 * 1)(4096)
    assert L.mnrf_dw_planes_workspace_floats(1, one) > 0
    assert L.mnrf_field_backward_planes(null, 4, null, 3, null, null, 1, null, null, null, null, null, null, null, null, null,
                                        null, null, null, null, null, 0, null) < 0
    offs = (C.c_int64 * 17)(*range(0, 17 * 1024, 1024))
    assert L.mnrf_tcnn_backward_workspace_floats2(offs, 16) == L.mnrf_tcnn_backward_workspace_floats(offs) + 16 * 1024 + 4      # (+ overflow word)
    assert L.mnrf_tcnn_backward_workspace_floats2(offs, 0) == L.mnrf_tcnn_backward_workspace_floats(offs)
    # zero-sized work is a no-op, not an error
    assert L.mnrf_field_composite_fused(null, 0, null, null, null, 27, 0, null, null, null, null, null, null, null, null) < 0   # (pointers are checked first)
    assert L.mnrf_embed(null, 0, 3, 4, null, null) == 0
    assert L.mnrf_threshold_mask(null, 0, null, null) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mirror_nerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensors_are_rejected():
    import torch
    import mirror_nerf_amd as M
    with pytest.raises(RuntimeError):
        M.render_rays({}, {"xyz": M.Embedding(10), "dir": M.Embedding(4)}, torch.zeros(4, 8))


def test_module_mirrors_reference_names():
    import torch
    import mirror_nerf_amd as M
    from mirror_nerf_amd.weights import PARAM_NAMES, PARAM_SHAPES
    from tests.golden import weights as GW
    torch.manual_seed(0)
    m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
    sd = m.state_dict()
    assert list(sd) == PARAM_NAMES
    ref = GW.make_state_dict(0, 1)[0]
    for k, v in sd.items():
        assert tuple(v.shape) == PARAM_SHAPES[k]
        assert (v.numpy() == ref[k]).all(), k     # same construction order => same init under a seed
    with pytest.raises(NotImplementedError):
        M.MirrorNeRF(W=128, predict_normal=True, predict_mirror_mask=True)
    with pytest.raises(NotImplementedError):
        M.MirrorNeRF(in_channels_xyz=75, in_channels_dir=27)      # Embedding(12): more bands than the kernels evaluate
    few = M.MirrorNeRF(in_channels_xyz=39, in_channels_dir=15, predict_normal=True, predict_mirror_mask=True)      # --N_emb_xyz 6 --N_emb_dir 2
    assert (few.n_freqs_xyz, few.n_freqs_dir) == (6, 2) and few.xyz_encoding_5[0].weight.shape == (256, 256 + 39)


def test_isa_invariants_of_the_built_field_kernels():
    """scripts/check_isa.py on the objects linked into libmnrf_hip.so: the counted s_waitcnt schemes of the field
    kernels require that no scalar load sits inside an MFMA range and that the forward kernels do not spill."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "ISA check ok" in r.stdout


def test_the_shipped_objects_were_built_with_the_default_flags():
    """csrc/.cxxflags records the flags of the objects in the tree (every object depends on it): an experiment's -DMNRF_EXP_*
    must never be what the default library was linked from, and the library must not be older than any of its objects."""
    csrc = os.path.join(ROOT, "mirror_nerf_amd", "csrc")
    flags = os.path.join(csrc, ".cxxflags")
    if not os.path.exists(flags):
        pytest.skip("library not built by this Makefile here")
    assert "MNRF_EXP" not in open(flags).read()
    lib = os.path.join(ROOT, "mirror_nerf_amd", "libmnrf_hip.so")
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o")]
    assert objs and all(os.path.getmtime(o) <= os.path.getmtime(lib) + 1.0 for o in objs)
    assert all(os.path.getmtime(o) >= os.path.getmtime(flags) - 1.0 for o in objs)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`--gpus N` must equal the launched world size (a mismatch used to run one process labelled n_gpus 1)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_gradient_scale_policy():
    """mirror_nerf._lower_gradient_scale: only a trip that is nothing but a scaled gradient of the training backward lowers the
    module's gradient scale (2^4 at a time, 2^8 at most); anything else leaves the decision to the fp32 fall-back."""
    import warnings
    import torch
    from mirror_nerf_amd import mirror_nerf as MN
    m = torch.nn.Linear(1, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert not MN._lower_gradient_scale(m, 1)                 # saturated, but not tagged as the backward
        assert not MN._lower_gradient_scale(m, 1 | 128 | 256)     # the forward saturated too
        assert not MN._lower_gradient_scale(m, 1 | 256 | 512)     # ... or the second-order pass
        assert not MN._lower_gradient_scale(m, 2 | 1 | 256)       # ... or a weight is out of range
        for want in (4, 8):
            assert MN._lower_gradient_scale(m, 1 | 256) and m.__dict__["_mnrf_seed_reduction"] == want
        assert not MN._lower_gradient_scale(m, 1 | 256)           # exhausted: fp32 from here
