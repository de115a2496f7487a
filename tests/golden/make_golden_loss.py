"""The scalar loss and gradient digest of fixture G9, shared by the generator (make_golden.py) and
the GPU test so that both sides evaluate the same formula."""
import numpy as np
import torch


def first_order_loss(res, target, gt):
    """A smooth scalar of the FIRST-ORDER outputs of the train-semantics render (everything except
    the keys derived from the normalised autograd gradient)."""
    c = torch.tensor([0.3, -0.5, 0.8], device=target.device)
    L = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean()
    L = L + 0.1 * ((res["mirror_mask_fine"] - gt) ** 2).mean() + 0.1 * ((res["mirror_mask_coarse"] - gt) ** 2).mean()
    L = L + 0.05 * res["depth_fine"].mean() + 0.05 * (res["x_surface_fine"] ** 2).mean()
    L = L + 0.1 * (res["surface_normal_fine"] * c).sum(-1).mean() + 0.01 * (res["weights_fine"] ** 2).sum(-1).mean()
    L = L + 0.02 * res["opacity_coarse"].mean()
    return L


def full_loss(res, target, gt):
    """first_order_loss + terms on the keys derived from the autograd normal (second-order gradients)."""
    c = torch.tensor([0.3, -0.5, 0.8], device=target.device)
    L = first_order_loss(res, target, gt)
    L = L + 0.05 * res["normal_dif_fine"].mean() + 0.05 * res["normal_dif_coarse"].mean()
    L = L + 0.02 * (res["surface_normal_grad_fine"] * c).sum(-1).mean()
    return L


def mask_focus_loss(res, target, gt):
    """A loss dominated by the composited mirror mask (fixtures G9b for the --detach_density_*_for_mask_loss options: with
    the density detached from this term, the trunk is only reached through the small colour term)."""
    L = 0.01 * (((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean())
    return L + ((res["mirror_mask_fine"] - gt) ** 2).mean() + ((res["mirror_mask_coarse"] - gt) ** 2).mean()


def grad_summary(t, like=None):
    if t is None:   # parameter not reached by the loss
        t = torch.zeros_like(like)
    g = t.detach().reshape(-1).double()
    k = min(g.numel(), 48)
    idx = torch.linspace(0, g.numel() - 1, k).long()
    return np.concatenate([[g.sum().item(), g.norm().item(), g.abs().max().item()], g[idx].numpy()])
