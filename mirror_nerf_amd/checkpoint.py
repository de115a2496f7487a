"""Checkpoint interchange with the reference (utils/__init__.py:109-136; train.py:56,66;
eval.py:996-1001): a Lightning `.ckpt` (or a plain dict) whose `state_dict` keys are
`nerf_coarse.<param>` / `nerf_fine.<param>` loads into our MirrorNeRF modules unchanged, and
`save_ckpt` writes the same layout back."""
import os
import pickle

import torch


def _load_file(path, trusted=False):
    """torch.load of a checkpoint file.  Tensors-only first (`weights_only=True`, the default of this torch).  Lightning
    1.5 checkpoints of the reference also pickle `hyper_parameters` (an argparse.Namespace / AttributeDict), callbacks
    and optimizer states: those load with the known plain containers allow-listed; anything else needs `trusted=True`
    (full unpickling executes code from the file -- only for checkpoints you wrote yourself)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        first = e
    import argparse
    import collections
    allow = [argparse.Namespace, collections.OrderedDict, collections.defaultdict, dict]
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if trusted:
            return torch.load(path, map_location="cpu", weights_only=False)
        raise RuntimeError(
            f"{path}: the checkpoint pickles objects outside the tensors-only allow-list ({e}); if you trust the file, "
            "pass trusted=True (load_ckpt / extract_model_state_dict) to unpickle it fully") from first


def extract_model_state_dict(ckpt, model_name="model", prefixes_to_ignore=(), trusted=False):
    """ckpt: path (str / bytes / os.PathLike) or an already loaded dict.  Returns {param name: tensor} of `model_name`."""
    checkpoint = _load_file(os.fspath(ckpt), trusted) if isinstance(ckpt, (str, bytes, os.PathLike)) else ckpt
    if "state_dict" in checkpoint:   # pytorch-lightning checkpoint
        checkpoint = checkpoint["state_dict"]
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def load_ckpt(model, ckpt, model_name="model", prefixes_to_ignore=(), trusted=False):
    if not ckpt:
        return
    sd = model.state_dict()
    got = extract_model_state_dict(ckpt, model_name, prefixes_to_ignore, trusted)
    assert len(got) > 0, f"the checkpoint holds no parameter named '{model_name}.*'"     # the reference asserts here too
    sd.update(got)
    model.load_state_dict(sd, strict=False)


def save_ckpt(path, system, epoch=0, global_step=0):
    """Writes {"state_dict": {"nerf_coarse.*", "nerf_fine.*"}, "epoch", "global_step"} like Lightning's
    ModelCheckpoint does for train.NeRFSystem."""
    sd = {}
    for name in ("nerf_coarse", "nerf_fine"):
        m = getattr(system, name, None)
        if m is not None:
            for k, v in m.state_dict().items():
                sd[f"{name}.{k}"] = v.detach().cpu()
    torch.save({"state_dict": sd, "epoch": epoch, "global_step": global_step}, path)
