"""Host logic of the training step that needs no GPU (round 6): the geometry stage's target masking (train.py:410-416) and the
contract between the loss function and the step (train.py:426, 439-446 hand both the same stage flag)."""
from types import SimpleNamespace

import pytest
import torch

from mirror_nerf_amd import training as T


def _reference_masking(hp, rgbs, mask, stage):
    """train.py:410-416 verbatim in behaviour: in place on a copy."""
    rgbs = rgbs.clone()
    if stage and not (mask < 0).any() and not hp.woMaskRGBtoBlack:
        rgbs[mask.bool()] = 0
    return rgbs


@pytest.mark.parametrize("wo_mask", [False, True])
@pytest.mark.parametrize("invalid", [False, True])
def test_stage_target_is_the_reference_s_masking(wo_mask, invalid):
    g = torch.Generator().manual_seed(3)
    rgbs = torch.rand(64, 3, generator=g)
    mask = (torch.rand(64, generator=g) < 0.3).float()
    if invalid:
        mask[5] = -1.0
    hp = SimpleNamespace(woMaskRGBtoBlack=wo_mask)
    want = _reference_masking(hp, rgbs, mask, True)
    for gt_valid in (None, not invalid):      # read from the data, or stated by the caller (static route)
        got = T.stage_target(hp, rgbs, mask, gt_valid)
        assert torch.equal(got, want)
    assert torch.equal(rgbs, rgbs.clone())      # (the input is left alone: the reference writes into the batch)


def test_total_loss_fn_carries_its_stage_and_takes_the_static_flag():
    fn = T.total_loss_fn(SimpleNamespace(use_plane_consistent_loss=True), epoch=2, train_geometry_stage=True)
    assert fn.needs_rays and fn.takes_static and fn.train_geometry_stage is True
    assert T.total_loss_fn().train_geometry_stage is False


def test_extra_info_hands_the_stage_to_the_renderer():
    hp = SimpleNamespace(only_one_field=False)
    ex = T.extra_info(hp, torch.zeros(4), epoch=3, train_geometry_stage=True)
    assert ex["train_geometry_stage"] is True and ex["current_epoch"] == 3 and ex["is_eval"] is False
