"""metrics.py of the reference (5-15) on the GPU: `mse` and `psnr` of a rendered frame against the ground truth without
moving either to the host (eval.py:804 computes PSNR on CPU copies of full frames).
`ssim` (kornia) is outside the hot path."""
import torch

from . import _lib


def _reduce(image_pred, image_gt, valid_mask):
    L = _lib.lib()
    a = image_pred.detach().float().contiguous()
    b = image_gt.detach().float().contiguous()
    if a.shape != b.shape:
        raise RuntimeError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    per, m = 1, None
    if valid_mask is not None:
        m = valid_mask.to(torch.uint8).contiguous()
        if m.shape == a.shape:
            per = 1
        elif m.shape == a.shape[:-1]:
            per = a.shape[-1]          # value[valid_mask] with a per-pixel mask keeps whole pixels
        else:
            raise RuntimeError("valid_mask must have the image's shape or its shape without the channel axis")
    out = torch.empty(3, dtype=torch.float32, device=a.device)
    part = torch.empty(2 * L.mnrf_mse_blocks(), dtype=torch.float32, device=a.device)
    _lib.check(L.mnrf_mse_psnr(_lib.ptr(a), _lib.ptr(b), None if m is None else m.data_ptr(), a.numel(), per,
                               _lib.ptr(part), _lib.ptr(out), _lib.stream()), "mnrf_mse_psnr")
    return out


def mse(image_pred, image_gt, valid_mask=None, reduction="mean"):
    if reduction != "mean":
        raise NotImplementedError("only reduction='mean' (what eval.py uses) runs on the device")
    return _reduce(image_pred, image_gt, valid_mask)[0]


def psnr(image_pred, image_gt, valid_mask=None, reduction="mean"):
    if reduction != "mean":
        raise NotImplementedError("only reduction='mean' (what eval.py uses) runs on the device")
    return _reduce(image_pred, image_gt, valid_mask)[1]
