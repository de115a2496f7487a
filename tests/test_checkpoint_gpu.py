"""Checkpoint interchange on the GPU (utils/__init__.py:109-136, train.py:56,66, eval.py:996-1001):
the trained pair of fixtures G11 written as a Lightning-shaped `.ckpt` (`nerf_coarse.*` / `nerf_fine.*` under
`state_dict`, plus `hyper_parameters`, `epoch`, optimizer states), loaded through `checkpoint.load_ckpt` into FRESH modules
the way eval.py does, packed for the HIP kernels and rendered -- against the reference's render of the same weights
(`g11_trained_render_test`, `g11_trained_eval_l2`), with the tolerances of the fixture tests."""
import argparse
import collections
import os

import numpy as np
import pytest
import torch

from tests.golden import fixtures as FX
from tests.test_hip_parity import DEV, _cmp_trained, _emb, _np

pytestmark = pytest.mark.gpu


def _write_ckpt(path, fx):
    z = np.load(os.path.join(FX.HERE, fx.meta["weights_file"]))
    sd = collections.OrderedDict()
    for mname in ("coarse", "fine"):
        for k in z.files:
            if k.startswith(mname + "__"):
                sd[f"nerf_{mname}.{k[len(mname) + 2:]}"] = torch.from_numpy(z[k].copy())
    assert len(sd) == 64
    sd["loss.dummy"] = torch.zeros(1)           # a key of another module: ignored by name
    torch.save({"state_dict": sd, "epoch": 29, "global_step": 28000, "pytorch-lightning_version": "1.5.4",
                "hyper_parameters": argparse.Namespace(N_samples=64, N_importance=128, model_type="nerf", predict_normal=True,
                                                       predict_mirror_mask=True),
                "callbacks": {"ModelCheckpoint": {"best_model_score": torch.tensor(19.1), "dirpath": "ckpts/exp"}},
                "optimizer_states": [{"state": {}, "param_groups": [{"lr": 5e-4, "betas": (0.9, 0.999), "params": list(range(64))}]}],
                "lr_schedulers": [{"last_epoch": 29}]}, path)


def _load_pair(path):
    import mirror_nerf_amd as M
    from mirror_nerf_amd import checkpoint as C
    models = {}
    for name in ("coarse", "fine"):                # eval.py:996-1001
        m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True).to(DEV)
        C.load_ckpt(m, path, model_name=f"nerf_{name}")
        models[name] = m.eval()
    return models


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_lightning_ckpt_to_packed_weights_to_render(tmp_path, precision):
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    fx = FX.Fixture("g11_trained_render_test")
    assert not fx.meta.get("tweaks"), "the plain trained pair"
    path = tmp_path / "epoch=29.ckpt"
    _write_ckpt(path, fx)
    old = MN.PRECISION
    MN.set_precision(precision)
    try:
        models = _load_pair(path)
        # the loaded parameters are the fixture's, bit for bit, and live on the GPU
        want = fx.state_dicts()
        for name, sd in zip(("coarse", "fine"), want):
            for k, v in models[name].state_dict().items():
                assert v.is_cuda and np.array_equal(v.cpu().numpy(), sd[k]), (name, k)
        m = fx.meta
        rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
        got = _np(M.render_rays(models, _emb(), rays, m["N_samples"], m["use_disp"], m["perturb"], m["noise_std"], m["N_importance"],
                                m["chunk"], m["white_back"], m["test_time"], **m["kwargs"]))
        _cmp_trained("ckpt->render_test", got, fx, list(FX.PER_SAMPLE_FINE) + ["normal_coarse", "normal_fine", "pred_normal_coarse"])
        # and through the eval recursion (two bounces)
        fe = FX.Fixture("g11_trained_eval_l2")
        me = fe.meta
        got = _np(M.batched_inference(models, _emb(), torch.from_numpy(fe.inputs["rays"]).to(DEV), me["N_samples"], me["N_importance"],
                                      False, me["chunk"], args=me["args"], trace_secondary_rays=True,
                                      normal_noise_std=me["args"]["normal_noise_std"]))
        _cmp_trained("ckpt->eval_l2", got, fe, FX.PER_SAMPLE_FINE)
    finally:
        MN.set_precision(old)


def test_reloading_other_weights_into_the_same_modules_is_seen(tmp_path):
    """load_ckpt goes through load_state_dict (in-place copy_): the packed image of a module that has already rendered
    must follow (weights.PackedCache keys on Tensor._version)."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd import checkpoint as C
    fx = FX.Fixture("g11_trained_render_test")
    path = tmp_path / "a.ckpt"
    _write_ckpt(path, fx)
    torch.manual_seed(5)
    models = {k: M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True).to(DEV)
              for k in ("coarse", "fine")}
    m = fx.meta
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)

    def render():
        return _np(M.render_rays(models, _emb(), rays, m["N_samples"], m["use_disp"], 0, 0, m["N_importance"], m["chunk"],
                                 m["white_back"], m["test_time"], **m["kwargs"]))
    before = render()                       # random-init weights: packs them
    for name in ("coarse", "fine"):
        C.load_ckpt(models[name], str(path), model_name=f"nerf_{name}")
    after = render()
    assert np.max(np.abs(before["rgb_fine"] - after["rgb_fine"])) > 1e-2
    _cmp_trained("reload", after, fx, list(FX.PER_SAMPLE_FINE) + ["normal_coarse", "normal_fine", "pred_normal_coarse"])
