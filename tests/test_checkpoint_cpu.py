"""Checkpoint interchange: Lightning-style state dicts of the reference load into
our modules by name, prefixes stripped, extra keys ignored."""
import torch

import mirror_nerf_amd as M
from mirror_nerf_amd import checkpoint as C
from tests.golden import weights as GW


def _mk():
    return M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)


def test_lightning_style_round_trip(tmp_path):
    sds = GW.make_state_dict(4, 2)
    ck = {"state_dict": {}, "epoch": 3}
    for name, sd in zip(("nerf_coarse", "nerf_fine"), sds):
        for k, v in sd.items():
            ck["state_dict"][f"{name}.{k}"] = torch.from_numpy(v)
    ck["state_dict"]["loss.weight"] = torch.zeros(1)          # ignored: other module
    path = tmp_path / "epoch=3.ckpt"
    torch.save(ck, path)
    coarse, fine = _mk(), _mk()
    C.load_ckpt(coarse, str(path), "nerf_coarse")
    C.load_ckpt(fine, str(path), "nerf_fine")
    for m, sd in ((coarse, sds[0]), (fine, sds[1])):
        for k, v in m.state_dict().items():
            assert (v.numpy() == sd[k]).all(), k
    # save in the same layout and load back
    system = torch.nn.Module()
    system.nerf_coarse, system.nerf_fine = coarse, fine
    C.save_ckpt(tmp_path / "last.ckpt", system, epoch=4)
    again = _mk()
    C.load_ckpt(again, str(tmp_path / "last.ckpt"), "nerf_fine", prefixes_to_ignore=["rgb"])
    assert (again.sigma.weight.detach().numpy() == sds[1]["sigma.weight"]).all()
    assert not (again.rgb[0].weight.detach().numpy() == sds[1]["rgb.0.weight"]).all()   # ignored prefix kept its init


def test_missing_model_name_asserts(tmp_path):
    import pytest
    with pytest.raises(AssertionError):
        C.load_ckpt(_mk(), {"state_dict": {"other.x": torch.zeros(1)}}, "nerf_coarse")


class _Opaque:      # stands for an arbitrary pickled object (a callback instance, a custom hparams class)
    def __init__(self):
        self.x = 1


def test_lightning_shaped_checkpoint_with_hyper_parameters(tmp_path):
    """A Lightning 1.5 .ckpt carries hyper_parameters (argparse.Namespace), callbacks and optimizer states besides the
    state_dict; torch >= 2.6 loads with weights_only=True by default.  Path objects are accepted."""
    import argparse
    import collections
    import pytest
    sd = GW.make_state_dict(4, 1)[0]
    ck = {"state_dict": collections.OrderedDict((f"nerf_coarse.{k}", torch.from_numpy(v)) for k, v in sd.items()),
          "epoch": 7, "global_step": 1234, "pytorch-lightning_version": "1.5.4",
          "hyper_parameters": argparse.Namespace(N_samples=64, N_importance=64, model_type="nerf"),
          "callbacks": {"ModelCheckpoint": {"best_model_score": torch.tensor(0.5), "dirpath": "ckpts/x"}},
          "optimizer_states": [{"state": {}, "param_groups": [{"lr": 5e-4, "betas": (0.9, 0.999), "params": [0, 1]}]}],
          "lr_schedulers": [{"last_epoch": 7, "milestones": collections.Counter({20: 1})}]}
    path = tmp_path / "epoch=7.ckpt"
    torch.save(ck, path)
    m = _mk()
    C.load_ckpt(m, path, "nerf_coarse")                       # pathlib.Path
    assert (m.sigma.weight.detach().numpy() == sd["sigma.weight"]).all()
    # an object outside the allow-list: refused with a clear message unless the caller trusts the file
    ck["callbacks"]["mine"] = _Opaque()
    torch.save(ck, path)
    with pytest.raises(RuntimeError, match="trusted=True"):
        C.load_ckpt(_mk(), path, "nerf_coarse")
    m2 = _mk()
    C.load_ckpt(m2, path, "nerf_coarse", trusted=True)
    assert (m2.sigma.weight.detach().numpy() == sd["sigma.weight"]).all()
