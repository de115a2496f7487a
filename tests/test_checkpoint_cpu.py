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
