"""A minimal training step around NeRFSystem: forward (train semantics), loss, backward through the
HIP kernels, one flat gradient all-reduce, Adam.  The Lightning loop of the reference
(train.py:386-458) is out of scope; this is the arithmetic a `training_step` performs, used by
bench.py and scripts/bench_train.py with synthetic batches (1024 rays, perturb = noise_std = 1,
train.py defaults)."""
import os
import time
from types import SimpleNamespace

import torch

from . import dist as D
from .recursion import NeRFSystem
from .weights import params_of


def default_hparams(**over):
    hp = dict(N_emb_xyz=10, N_emb_dir=4, predict_normal=True, predict_mirror_mask=True, model_type="nerf",
              N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=1.0, chunk=32768,
              trace_secondary_rays=True, only_one_field=False, max_recursive_level=1,
              only_trace_rays_in_mirrors=True, for_vis=False)
    hp.update(over)
    return SimpleNamespace(**hp)


def color_mask_loss_torch(res, target, gt_mask):
    """ColorLoss + MirrorMaskLoss of losses.py (7-51, 175-198) in their default form, with torch ops (~35 small launches forward
    and backward; kept as the yardstick of color_mask_loss)."""
    loss = ((res["rgb_coarse"] - target) ** 2).mean()
    if "rgb_fine" in res:
        loss = loss + ((res["rgb_fine"] - target) ** 2).mean()
    key = "mirror_mask_fine" if "mirror_mask_fine" in res else "mirror_mask_coarse"
    m = res[key].clamp(1e-5, 1 - 1e-5)
    return loss + 0.1 * torch.nn.functional.binary_cross_entropy(m, gt_mask)


_COLOR_MASK = []
_COLOR_MASK_RAYS = {}      # (n, device) -> the (n, 8) zero tensor the fused loss takes as `rays` (read by the absent normal terms only)


def _backward_unit(loss):
    """loss.backward() with the constant 1.0 of losses.unit_gradient as the root gradient (no fill launch for the root, no scaling
    launch inside the fused loss); any loss that is not a float32 CUDA scalar takes the plain call."""
    if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
        from .losses import unit_gradient
        loss.backward(gradient=unit_gradient(loss.device))
    else:
        loss.backward()


def color_mask_loss(res, target, gt_mask):
    """ColorLoss + MirrorMaskLoss of losses.py (7-51, 175-198) in their default form: the reference's TotalLoss restricted to the
    two terms that read no normal_* key (so the backward has no second-order pass), evaluated by the fused loss kernel
    (mirror_nerf_amd/losses.py: value AND gradients in one launch) instead of ~35 torch launches."""
    if not _COLOR_MASK:
        from .losses import get_loss
        _COLOR_MASK.append(get_loss(SimpleNamespace()))
    inputs = {k: res[k] for k in ("rgb_coarse", "rgb_fine", "mirror_mask_coarse", "mirror_mask_fine") if k in res}
    key = (target.shape[0], str(target.device))
    zr = _COLOR_MASK_RAYS.get(key)
    if zr is None:
        zr = _COLOR_MASK_RAYS[key] = torch.zeros(target.shape[0], 8, device=target.device)
    return _COLOR_MASK[0](inputs, {"rgbs": target, "mirror_mask": gt_mask, "rays": zr}, train_geometry_stage=False, epoch=5)[0]


def total_loss_fn(hparams=None, epoch=5, train_geometry_stage=False):
    """The reference's training loss (train.py:439-446): losses.TotalLoss on the result dict, evaluated by the fused HIP
    loss kernels (mirror_nerf_amd/losses.py).  With the default hparams and epoch >= 1 it reads normal_dif_* and
    normal_fine, so the backward includes the second-order pass through the density-gradient normal.  With
    hparams.use_plane_consistent_loss (run.sh:277) the plane term draws its quadruples on the device when the step is static or
    being captured (`static=True`: no host read of the mirror-ray count), from the CPU generator like the reference otherwise."""
    from .losses import get_loss
    crit = get_loss(hparams if hparams is not None else SimpleNamespace())

    def fn(res, target, gt_mask, rays, static=False):
        batch = {"rgbs": target, "mirror_mask": gt_mask, "rays": rays}
        if static:
            batch["_plane_on_device"] = True
        return crit(res, batch, train_geometry_stage=train_geometry_stage, epoch=epoch)[0]
    fn.needs_rays = True
    fn.takes_static = True
    fn.train_geometry_stage = bool(train_geometry_stage)
    return fn


def _call_loss(loss_fn, res, target, gt_mask, rays, static):
    if getattr(loss_fn, "takes_static", False):
        return loss_fn(res, target, gt_mask, rays, static=static)
    return loss_fn(res, target, gt_mask, rays) if getattr(loss_fn, "needs_rays", False) else loss_fn(res, target, gt_mask)


def stage_target(hparams, target, gt_mask, gt_valid):
    """train.py:410-416 -- in the geometry stage the colour target inside the (valid) ground-truth mirror mask is black, unless
    --woMaskRGBtoBlack leaves those pixels out of the colour loss instead.  gt_valid None: read from the device as the reference does."""
    if getattr(hparams, "woMaskRGBtoBlack", False):
        return target
    if gt_valid is None:
        gt_valid = not bool((gt_mask < 0).any().item())
    if not gt_valid:
        return target
    return torch.where((gt_mask != 0).view(-1, 1), torch.zeros((), dtype=target.dtype, device=target.device), target)


def static_step_ok(system):
    """Can `system` take the static training route (recursion.render_rays_chunk_recursively "STATIC STEP": the reflected-ray
    count never visits the host)?  MirrorNeRF fields on the split arithmetic with operand planes, or -- round 6 -- hash-grid fields
    (mnrf_tcnn_forward_n / _backward_n); a model the range guard has pinned to fp32 and MNRF_DW_PLANES=0 use the host-driven route."""
    from . import autograd as AG
    from .mirror_nerf import MirrorNeRF, precision_of
    from .mirror_nerf_tcnn import MirrorNeRFTcnn
    models = list(system.models.values())
    if models and all(isinstance(m, MirrorNeRFTcnn) for m in models):
        return True      # round 6: the hash-grid field takes the live count too (mnrf_tcnn_forward_n / _backward_n)
    return AG.DW_PLANES and all(isinstance(m, MirrorNeRF) and precision_of(m) == "split" for m in models)


# MNRF_STATIC_STEP=0: train_step keeps the host-driven route (one device->host read of the reflected-ray count per level and step,
# like train.py:175) even when the caller states `gt_valid`
STATIC_STEP = os.environ.get("MNRF_STATIC_STEP", "1") != "0"


def extra_info(hp, gt_mask, epoch=0, train_geometry_stage=False):
    """The `extra` dict training_step hands to NeRFSystem.forward (train.py:420-436), incl. the gradient-steering options."""
    return {"mirror_mask": gt_mask, "is_eval": False, "train_geometry_stage": train_geometry_stage,
            "only_one_field": getattr(hp, "only_one_field", False),
            "only_one_field_fine_epoch": getattr(hp, "only_one_field_fine_epoch", 2), "current_epoch": epoch,
            "detach_density_outside_mirror_for_mask_loss": getattr(hp, "detach_density_outside_mirror_for_mask_loss", False),
            "detach_density_for_mask_loss": getattr(hp, "detach_density_for_mask_loss", False),
            "detach_density_for_normal_loss": getattr(hp, "detach_density_for_normal_loss", False)}


# What a training step does when the range guard of the split arithmetic trips (mirror_nerf.check_guard):
#   "skip"  (default) the optimizer update of the tripping step is SKIPPED ON THE DEVICE -- torch's fused Adam takes the guard
#           flag as its `found_inf` (the GradScaler mechanism), all-reduced (MAX) over the ranks first, so every rank skips the
#           same step -- with no host read: nothing computed from saturated operands ever reaches the weights, the queue is
#           not drained, and one batch is lost.  The host learns it during the next step and pins the models to the fp32 kernels.
#   "sync"  (MNRF_GUARD_SYNC=1 / MNRF_GUARD_MODE=sync; also the fall-back for optimizers without `found_inf`) the flag is
#           read before the optimizer step (a queue drain, ~0.7 ms): a tripped step is recomputed on the fp32 kernels.
# (The round-2/3 "async" mode -- the tripping step's update applied, a warning one step late -- is gone: round 6.)
GUARD_MODE = os.environ.get("MNRF_GUARD_MODE") or ("sync" if os.environ.get("MNRF_GUARD_SYNC", "0") == "1" else "skip")
if GUARD_MODE not in ("skip", "sync"):
    raise ValueError(f"MNRF_GUARD_MODE={GUARD_MODE!r}: 'skip' or 'sync'")


_SETTLE_LATE = os.environ.get("MNRF_GUARD_SETTLE_LATE", "1") != "0"      # 0: read the previous step's flags at the start of a step (A/B)


_STATE_FLAGS = {}


def _state_flags(device, has_split, has_pinned):
    """[0, has_split, has_pinned] on `device`, built once per combination (a torch.tensor(list, device=cuda) per step is a
    synchronous host-to-device copy: it drains the queue)."""
    key = (str(device), bool(has_split), bool(has_pinned))
    t = _STATE_FLAGS.get(key)
    if t is None:
        t = _STATE_FLAGS[key] = torch.tensor([0.0, float(has_split), float(has_pinned)], device=device)
    return t


def _takes_found_inf(optimizer):
    return bool(getattr(optimizer, "defaults", {}).get("fused")) or hasattr(optimizer, "mnrf_found_inf")


def train_step(system, optimizer, rays, target, gt_mask, loss_fn=color_mask_loss, epoch=0, gt_valid=None):
    """One optimisation step: forward (train semantics), loss, backward, gradient all-reduce, optimizer.  See GUARD_MODE for
    what happens when the split-f16 arithmetic leaves its range during the step.
    Geometry stage (run.sh:276 --train_geometry_stage; train.py:386-416, 426): `system.train_geometry_stage` as the reference's
    module attribute -- no reflections are traced, the target inside the mirror is black (stage_target); the loss function built by
    total_loss_fn(..., train_geometry_stage=True) carries the stage's flag (train.py:439-446).
    gt_valid: the caller's statement that every entry of gt_mask is valid (>= 0; True) or that some are not (False) -- what
    train.py:153 reads from the device.  With it (and MNRF_STATIC_STEP != 0, MirrorNeRF fields on the split arithmetic) the step
    takes the STATIC route: no device->host read at all, the reflected-ray count stays on the device (recursion.py)."""
    from .mirror_nerf import check_guard, guard_async_begin, guard_async_end, pin_fp32
    rank, world = D.world()
    collective = world > 1 or D.forced()
    mode = GUARD_MODE if (GUARD_MODE == "sync" or _takes_found_inf(optimizer)) else "sync"
    token = system.__dict__.pop("_mnrf_guard_token", None)

    def settle(token):          # the previous step's flags
        tok, flags_host, tmode, ev = token
        tripped_here = guard_async_end(tok, adapt=flags_host is None)      # (one rank, update skipped)
        tripped_any, mixed = tripped_here, False
        if flags_host is not None:      # more than one rank: [a rank tripped, a rank runs split models, a rank runs pinned ones]
            if ev is not None:
                ev.synchronize()
            f = flags_host.tolist()
            tripped_any, mixed = f[0] != 0.0, (f[1] != 0.0 and f[2] != 0.0)
            if tripped_any or mixed:
                pin_fp32(system)        # EVERY rank pins EVERY model, whoever tripped (or was pinned outside train_step: a validation
                                        # pass on one rank) -- the ranks stay on the same kernels
        if tripped_any or tripped_here:
            import warnings
            warnings.warn("mirror_nerf_amd: the previous training step left the range of the split-f16 arithmetic; "
                          "its optimizer update was skipped on every rank; the models run on the fp32 kernels from now on",
                          RuntimeWarning, stacklevel=3)
    # Reading them HERE would wait for the device to finish the previous step's backward pass (their event sits behind it) and
    # the host could not queue this step's primary pass ahead of time: the device then idles ~0.1 ms at every step start while
    # the first launches arrive.  On one rank the flags are settled after this step's forward has been queued -- the host waits
    # there anyway (the reflected-ray count) -- unless they have arrived already; with more than one rank they are settled
    # now (the pinning decision must reach every rank before any of them issues this step's collectives).
    late = _SETTLE_LATE and token is not None and token[0] is not None and not collective and mode == "skip" and not token[0][2].query()
    if token is not None and not late:
        settle(token)

    static = STATIC_STEP and gt_valid is not None and static_step_ok(system)

    stage = bool(getattr(system, "train_geometry_stage", False))
    if stage:
        target = stage_target(system.hparams, target, gt_mask, gt_valid)

    def fwd_bwd():
        nonlocal late
        ex = dict(extra_info(system.hparams, gt_mask, epoch, stage), _guard=False)
        if static and static_step_ok(system):      # (re-checked: a recomputed step runs after a trip has pinned the models)
            ex.update(_static=True, _gt_valid=bool(gt_valid))
        res = system(rays, ex)
        if late and not static:  # (a trip pins the models for the NEXT forward; this step's own flags gate its own update)
            settle(token)
            late = False
        loss = _call_loss(loss_fn, res, target, gt_mask, rays, bool(ex.get("_static")))
        optimizer.zero_grad(set_to_none=True)
        _backward_unit(loss)
        return loss
    loss = fwd_bwd()
    if collective:
        D.issue_pending()       # the guard's own all-reduce below must come after ALL buckets on every rank (readiness inside backward is rank-local)
    if mode == "sync":
        tripped = check_guard(system)          # (pins the tripping models to fp32)
        if collective:                         # the decision must be the same on every rank: they all recompute, or none
            from .mirror_nerf import MirrorNeRF, precision_of
            fields = [m for m in system.models.values() if isinstance(m, MirrorNeRF)]
            flag = torch.tensor([1.0 if tripped else 0.0,      # + the ranks' states, as in skip mode: ranks pinned outside the step
                                 float(any(precision_of(m).startswith("split") for m in fields)),
                                 float(any(not precision_of(m).startswith("split") for m in fields))], device=rays.device)
            if D._TRACE:
                D._trace("guard flags (sync mode)")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            f = flag.tolist()
            if f[0] != 0.0:
                pin_fp32(system)               # every rank, every model: the ranks stay on the same kernels
                tripped = True
            elif f[1] != 0.0 and f[2] != 0.0:
                pin_fp32(system)               # ranks in different states (one was pinned by a validation pass): converge; the
                                               # gradients of this pass are valid on either arithmetic, nothing is recomputed
        if tripped:
            D.reset_overlap()                  # the first pass's bucket all-reduces (issued from inside backward) are discarded
            loss = fwd_bwd()                   # this step again, exactly
            D.issue_pending()
            from .mirror_nerf import release_transient
            release_transient(system)          # (one rank, a range-only trip: the batch's rays, not the model -- back to split)
    else:
        tok = guard_async_begin(system)        # forward + backward flags of THIS step, read at the next (None: no split model here)
        flags_host = flags_ev = found = None
        if tok is not None:
            found = (tok[3] != 0).any().to(torch.float32)          # 0-dim, like GradScaler's found_inf
        if collective:
            # ALWAYS issued, whatever this rank's own state: precision pinning outside train_step is rank-local -- a
            # validation pass that trips the guard on rank 0 pins rank 0 only -- and a rank that skipped this collective because it
            # has no split model left would meet the others' flag with its next bucket.  Three MAX-reduced flags: [a rank tripped
            # in this step | a rank runs split models | a rank runs pinned models]; a trip, or ranks in different states, pins
            # every model on every rank at the next step's start (settle).
            from .mirror_nerf import MirrorNeRF, precision_of
            fields = [m for m in system.models.values() if isinstance(m, MirrorNeRF)]
            has_split = any(precision_of(m).startswith("split") for m in fields)
            has_pinned = any(not precision_of(m).startswith("split") for m in fields)
            vec = _state_flags(rays.device, has_split, has_pinned).clone()
            if tok is not None:
                vec[0] = (tok[3] != 0).any()
            if D._TRACE:
                D._trace("guard flags (skip mode)")
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.MAX)
            flags_host = torch.empty(3, dtype=torch.float32, pin_memory=rays.is_cuda)
            flags_host.copy_(vec, non_blocking=True)
            if rays.is_cuda:
                flags_ev = torch.cuda.Event()
                flags_ev.record()
            found = vec[0]
        # fused Adam: no update where found_inf != 0.  None once no split model is left anywhere: the flag tensor of a tripping
        # step must not keep vetoing updates (found with two ranks, where a trip pins EVERY model)
        optimizer.grad_scale, optimizer.found_inf = None, found
        system.__dict__["_mnrf_guard_token"] = (tok, flags_host, mode, flags_ev)
    # RCCL over xGMI when world_size > 1: per-model flat buckets, the all-reduce of a model issued from inside the backward
    # pass as soon as its gradients are complete (dist.attach_overlap), only waited for here
    D.allreduce_gradients(params_of(system), modules=list(system.models.values()))
    optimizer.step()
    if late:                    # static route: the host never waits inside the step; the previous step's flags are read here, with
        settle(token)           # this whole step queued (a trip then costs the batch of this step too: its flags trip as well)
    return loss


class FlatAdam:
    """Opt-in optimizer for field models (MirrorNeRF with all 32 parameters at their full shapes): Adam over ONE flat parameter
    tensor per model instead of 32 -- the parameters become views of it, and its gradient IS the flat buffer the backward pass
    already produced (autograd._Pending: the `.grad`s of the parameters are views of one buffer), so nothing is gathered or
    copied.  The update is torch.optim.Adam's (same arithmetic, same `found_inf` contract: train_step's range guard) from
    `mnrf_adam_step`, one thread per four elements: torch's fused multi-tensor kernel deals 65 536-element chunks to blocks, TEN
    blocks for a model, 46 us per model and step against 6 (round 4).  `kernel=False` (or MNRF_FLAT_ADAM_KERNEL=0) keeps torch's
    fused Adam over the flat tensors.  Build it AFTER moving the models to their device (step() raises when a parameter no longer
    aliases its flat tensor).  Not a torch.optim.Optimizer: lr_scheduler constructors reject it -- schedule by writing
    `param_groups[0]["lr"]` (read at every step; GraphedTrainStep copies it to the device between replays)."""

    def __init__(self, modules, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, kernel=None):
        from .weights import param_refs
        self.modules, self.flats = list(modules), []
        for m in self.modules:
            lay, total = D._field_layout(m)
            refs = param_refs(m)
            if lay is None or len(refs) != len(lay) or any(sub._parameters[pn].numel() != lay[full][1] for sub, pn, full in refs):
                raise ValueError("FlatAdam: needs MirrorNeRF modules with all 32 parameters at their full shapes")
            dev = refs[0][0]._parameters[refs[0][1]].device
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            for sub, pn, full in refs:
                q = sub._parameters[pn]
                o, k = lay[full]
                flat[o:o + k].copy_(q.data.reshape(-1))
                q.data = flat[o:o + k].view(q.shape)
            self.flats.append(torch.nn.Parameter(flat))
        self.kernel = (os.environ.get("MNRF_FLAT_ADAM_KERNEL", "1") != "0") if kernel is None else bool(kernel)
        self.inner = torch.optim.Adam(self.flats, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self.defaults = dict(self.inner.defaults)
        self.param_groups = self.inner.param_groups      # (lr / betas / eps / weight_decay are read from here at every step)
        if self.kernel:
            self._m = [torch.zeros_like(fp.data) for fp in self.flats]
            self._v = [torch.zeros_like(fp.data) for fp in self.flats]
            self._skipped = [torch.zeros(1, dtype=torch.int32, device=fp.device) for fp in self.flats]
            self._calls = 0
            self._steps = [0] * len(self.flats)      # per-module step counts, as torch.optim.Adam keeps per-parameter ones: a module
                                                     # that sat out k steps (no gradient) resumes with ITS count's bias corrections
            self._gs = self._fi = None

    # train_step hands the guard flag over as GradScaler does
    def _get(self, name):
        return getattr(self, "_gs" if name == "grad_scale" else "_fi") if self.kernel else getattr(self.inner, name, None)

    def _set(self, name, v):
        if self.kernel:
            setattr(self, "_gs" if name == "grad_scale" else "_fi", v)
        else:
            setattr(self.inner, name, v)
    grad_scale = property(lambda self: self._get("grad_scale"), lambda self, v: self._set("grad_scale", v))
    found_inf = property(lambda self: self._get("found_inf"), lambda self, v: self._set("found_inf", v))

    def zero_grad(self, set_to_none=True):
        from .weights import params_of
        for m, fp in zip(self.modules, self.flats):
            fp.grad = None
            for q in params_of(m):
                q.grad = None

    def _check_aliasing(self):
        """The parameters must still BE views of the flat tensors: `system.to()`, `.float()`, `load_state_dict(assign=True)` replace a
        parameter's storage, and step() would then update an orphaned flat tensor while training stood still without any error
        (ADVICE r4).  Checked on the first and last parameter of every module, two pointer compares per step."""
        from .weights import param_refs
        # (every parameter on the first call and every 16th, the first and the last one -- what a move or a cast changes -- always)
        n_chk = self.__dict__["_alias_checks"] = self.__dict__.get("_alias_checks", 0) + 1
        for m, fp in zip(self.modules, self.flats):
            lay, _total = D._field_layout(m)
            refs = param_refs(m)
            for sub, pn, full in (refs if n_chk % 16 == 1 else (refs[0], refs[-1])):
                q = sub._parameters[pn]
                if q.data_ptr() != fp.data_ptr() + 4 * lay[full][0]:
                    raise RuntimeError(f"FlatAdam: parameter {full} no longer aliases its flat tensor (the module was moved, cast or "
                                       "re-assigned after the optimizer was built): build FlatAdam again")

    def step(self):
        self._check_aliasing()
        from .weights import params_of
        grads = []
        for m, fp in zip(self.modules, self.flats):
            if all(q.grad is None for q in params_of(m)):
                grads.append(None)                   # no backward pass reached this module: torch.optim.Adam skips such parameters
                fp.grad = None                       # (an update from a zero gradient would still move them by momentum)
                continue
            flat = D._flat_bucket(m)                 # the backward pass's buffer when every .grad still is a view of it ...
            if flat is None:
                flat, _copied = D._module_message(m)   # ... else a flat copy in the same layout (zeros where there is no gradient)
            fp.grad = flat
            grads.append(flat)
        if not self.kernel:
            self.inner.step()
            return
        from . import _lib
        from .weights import bump_generation
        L, p = _lib.lib(), _lib.ptr
        self._calls += 1
        f32 = lambda t: None if t is None else t.to(torch.float32).reshape(-1)  # noqa: E731
        gs, fi = f32(self._gs), f32(self._fi)
        for i, (fp, g) in enumerate(zip(self.flats, grads)):
            if g is None:
                continue
            grp = self.param_groups[0]
            self._steps[i] += 1
            _lib.check(L.mnrf_adam_step(p(fp.data), p(g.contiguous()), p(self._m[i]), p(self._v[i]), fp.numel(), float(grp["lr"]),
                                        float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]),
                                        self._steps[i], p(self._skipped[i]), p(gs), p(fi), _lib.stream()), "mnrf_adam_step")
        self._gs = self._fi = None                   # (GradScaler semantics: the flags belong to one step)
        bump_generation()                            # the packed weight images are stale now (torch optimizers do this through a hook)

    # ---- hyper-parameters and the step count in device memory: what a step captured in a hipGraph needs (GraphedTrainStep)
    def enable_device_state(self):
        if not self.kernel:
            raise RuntimeError("FlatAdam(kernel=False) has no device-resident state")
        dev = self.flats[0].device
        self._hyper_host = None
        self._hyper = torch.zeros(5, dtype=torch.float64, device=dev)
        self._step_dev = torch.tensor([self._calls], dtype=torch.int64, device=dev)
        self._state_dev = torch.zeros(8, dtype=torch.float32, device=dev)      # the scalars of a step (mnrf_adam_prep)
        self.sync_hyper()

    def sync_hyper(self):
        """param_groups[0] -> device (call between steps: an lr scheduler changes the group's lr on the host)."""
        grp = self.param_groups[0]
        h = (float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]))
        if h != self._hyper_host:
            self._hyper.copy_(torch.tensor(h, dtype=torch.float64))
            self._hyper_host = h

    def step_dev(self, guard_words=()):
        """step() with every scalar read from device memory: capturable; the host's call count is NOT advanced (replays do not
        pass through here): GraphedTrainStep keeps it in step with the device count."""
        from . import _lib
        self._check_aliasing()
        L, p = _lib.lib(), _lib.ptr
        f32 = lambda t: None if t is None else t.to(torch.float32).reshape(-1)  # noqa: E731
        gs, fi = f32(self._gs), f32(self._fi)
        import ctypes
        gw = (ctypes.c_void_p * max(1, len(guard_words)))(*[w.data_ptr() for w in guard_words]) if guard_words else None
        _lib.check(L.mnrf_adam_prep(self._hyper.data_ptr(), self._step_dev.data_ptr(), p(self._skipped[0]), p(gs), p(fi), gw,
                                    len(guard_words), p(self._state_dev), _lib.stream()), "mnrf_adam_prep")
        from .weights import params_of
        grads, live = [], []
        for i, (m, fp) in enumerate(zip(self.modules, self.flats)):
            if all(q.grad is None for q in params_of(m)):
                grads.append(None)         # as step(): a module no backward pass reached is left alone (decided when the step is
                fp.grad = None             # captured; the device step count is shared, so such a module must sit out for good)
                continue
            flat = D._flat_bucket(m)
            if flat is None:
                flat, _copied = D._module_message(m)
            fp.grad = flat
            grads.append(flat.contiguous())
            live.append(i)
        for i in live:
            if self._steps[i] != self._calls:
                raise RuntimeError("FlatAdam.step_dev: the device-resident step count is shared by the modules of a step, but module "
                                   f"{i} has taken {self._steps[i]} of {self._calls} steps (it sat out earlier): use step()")
        self._live_dev = live
        for i0 in range(0, len(live), 4):  # all models of the step in one launch (mnrf_adam_step_dev_n takes up to four tensors)
            idx = live[i0:i0 + 4]
            vp = lambda ts: (ctypes.c_void_p * len(idx))(*[t.data_ptr() for t in ts])  # noqa: E731
            _lib.check(L.mnrf_adam_step_dev_n(len(idx), vp([self.flats[i].data for i in idx]), vp([grads[i] for i in idx]),
                                              vp([self._m[i] for i in idx]), vp([self._v[i] for i in idx]),
                                              (ctypes.c_int64 * len(idx))(*[self.flats[i].numel() for i in idx]), p(self._state_dev),
                                              vp([self._skipped[i] for i in idx]), _lib.stream()), "mnrf_adam_step_dev_n")
        self._gs = self._fi = None

    def state_dict(self):
        if not self.kernel:
            return self.inner.state_dict()
        return {"kernel": True, "calls": self._calls, "steps": list(self._steps),
                "exp_avg": [t.clone() for t in self._m], "exp_avg_sq": [t.clone() for t in self._v],
                "skipped": [t.clone() for t in self._skipped], "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                                for g in self.param_groups]}

    def load_state_dict(self, sd):
        if not self.kernel:
            return self.inner.load_state_dict(sd)
        self._calls = int(sd["calls"])
        self._steps = [int(v) for v in sd.get("steps", [self._calls] * len(self.flats))]
        for dst, src in zip(self._m + self._v + self._skipped, sd["exp_avg"] + sd["exp_avg_sq"] + sd["skipped"]):
            dst.copy_(src)
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)


class GraphedTrainStep:
    """A whole optimisation step -- weight packing, forward with reflections (static route), loss, backward, weight-gradient
    GEMMs, gradient all-reduce, Adam -- captured ONCE as a hipGraph (torch.cuda.CUDAGraph) and replayed per batch: no host work,
    no device->host read, no launch gaps.  What makes it possible is the static route of the recursion (the reflected-ray count
    stays on the device, every nested launch is sized for the batch and takes the count as its live row count: include/mnrf.h
    "live row counts on the device") and FlatAdam's device-resident step count and hyper-parameters.  The reference's step is
    train.py:386-458 (`training_step` + Lightning's backward / optimizer step); reference behaviour kept: train.py:153-178, 248-296.

        step = GraphedTrainStep(system, FlatAdam(...), batch=1024, gt_valid=True)
        loss = step(rays, target, gt_mask)        # tensors of the captured shapes; `loss` is a static tensor (read it late)

    The loss may be the whole of run.sh:259-280's recipe: total_loss_fn(hparams with use_plane_consistent_loss, epoch,
    train_geometry_stage) -- under capture the plane term draws on the device (losses.py); `system.train_geometry_stage` and the
    epoch are baked into the graph: set_epoch() / a changed stage captures again.

    More than one rank (round 6; DDP semantics of train.py:577-584): the collectives of the step are INSIDE the graph --
    the per-model bucket all-reduces issued from the backward hooks (dist.attach_overlap), and the range guard's words OR-ed over
    the ranks (RCCL has no bitwise reduction: one int32 lane per flag bit, MAX) so that every rank vetoes the same update on the
    device and makes the same decision one step late on the host.  Proven on one GPU with MNRF_FORCE_COLLECTIVES=1 (RCCL, world
    size 1: tests/test_dist_gpu.py); when the capture of a collective fails on a stack, `capture_error` says why and the step
    continues on train_step's static route (same launches, host-issued).

    Range guard: the guard words of the step gate the update ON THE DEVICE (found_inf) and are read by the host up to two replays
    late (the host queues replay N+1 before replay N has finished; it only waits when it is two ahead); a trip ends the graph --
    the models continue on train_step (pinned to fp32) -- or, for a gradient-scale trip, the step is captured again with the lower
    scale.  Precision pinning OUTSIDE the step (a validation pass that trips on one rank) does not reach a captured graph: its
    launches are baked, and its own guard words keep watching them."""

    _BITS = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048)

    def __init__(self, system, optimizer, batch, loss_fn=None, epoch=0, gt_valid=True, warmup=2):
        if not isinstance(optimizer, FlatAdam) or not optimizer.kernel:
            raise ValueError("GraphedTrainStep needs training.FlatAdam (its kernel form)")
        if not static_step_ok(system):
            raise RuntimeError("GraphedTrainStep: MirrorNeRF fields on the split arithmetic with operand planes only")
        self.system, self.opt, self.loss_fn, self.epoch, self.gt_valid = system, optimizer, loss_fn or color_mask_loss, epoch, bool(gt_valid)
        self._check_stage()
        self.collective = D.world()[1] > 1 or D.forced()
        if self.collective and torch.distributed.get_backend() != "nccl":
            raise NotImplementedError("GraphedTrainStep across ranks needs the RCCL backend (\"nccl\"): gloo stages through the host")
        dev = optimizer.flats[0].device
        self.rays = torch.zeros(batch, 8, device=dev)
        self.target = torch.zeros(batch, 3, device=dev)
        self.gt = torch.zeros(batch, device=dev)
        self.graph, self.loss, self.ended, self.capture_error = None, None, False, None
        import collections
        self._pending = collections.deque()       # (event, pinned words, gradient-scale reductions at issue) per replay not yet settled
        self._hosts, self._replays = None, 0
        self._stage_captured = None
        self._bits = torch.tensor(self._BITS, dtype=torch.int32, device=dev) if self.collective else None
        self.warmup = warmup
        if not hasattr(optimizer, "_step_dev"):
            optimizer.enable_device_state()

    def _stage(self):
        return bool(getattr(self.system, "train_geometry_stage", False))

    def _check_stage(self):
        want = getattr(self.loss_fn, "train_geometry_stage", None)
        if want is not None and want != self._stage():
            raise ValueError(f"GraphedTrainStep: the loss was built for train_geometry_stage={want} but system.train_geometry_stage is "
                             f"{self._stage()} (train.py:426, 439-446 hand both the same flag)")

    def set_epoch(self, epoch, loss_fn=None):
        """The epoch (and the loss built for it) are baked into the graph: a change captures again at the next call."""
        if epoch != self.epoch or (loss_fn is not None and loss_fn is not self.loss_fn):
            self.epoch, self.graph = epoch, None
            if loss_fn is not None:
                self.loss_fn = loss_fn
            self._check_stage()

    def _body(self):
        from .weights import invalidate_packed
        models = list(self.system.models.values())
        for m in models:
            invalidate_packed(m)          # the packing launches belong to every replay: the weights changed in the step before
        stage = self._stage()
        target = stage_target(self.system.hparams, self.target, self.gt, self.gt_valid) if stage else self.target
        ex = dict(extra_info(self.system.hparams, self.gt, self.epoch, stage), _guard=False, _static=True, _gt_valid=self.gt_valid)
        res = self.system(self.rays, ex)
        loss = _call_loss(self.loss_fn, res, target, self.gt, self.rays, True)
        self.opt.zero_grad(set_to_none=True)
        _backward_unit(loss)
        # a saturated step must not reach the weights: the guard words of the models' packed images veto the update inside the
        # Adam kernel (no torch ops to form a found_inf tensor)
        words = [m.__dict__["_mnrf_packed"].packed[-1:].view(torch.int32) for m in models]
        if self.collective:
            D.issue_pending()             # every bucket first, on every rank (readiness inside backward is rank-local), then the words
            lanes = torch.cat(words).view(-1, 1) & self._bits          # (models, 12): 0 or the bit
            if D._TRACE:
                D._trace("guard words (captured step)")
            torch.distributed.all_reduce(lanes, op=torch.distributed.ReduceOp.MAX)
            self._words_all = lanes.sum(1, dtype=torch.int32)          # the OR over the ranks: the same words everywhere
            words = [self._words_all[i:i + 1] for i in range(len(models))]
            D.allreduce_gradients(params_of(self.system), modules=models)
        self._word_views = words
        self.opt.step_dev(words)
        return loss.detach()

    def capture(self):
        opt = self.opt
        self._check_stage()
        keep = [t.clone() for t in opt.flats] + [t.clone() for t in opt._m + opt._v + opt._skipped] + [opt._step_dev.clone()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):     # warm-up off the capture: lazy initialisations, allocator, autograd threads
            for _ in range(self.warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():             # the warm-up steps were rehearsals: weights and optimizer state as before
            for dst, src in zip([t.data for t in opt.flats] + opt._m + opt._v + opt._skipped + [opt._step_dev], keep):
                dst.copy_(src)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        try:
            # With a process group alive its watchdog thread polls the events of earlier collectives (the warm-up's) -- an "unsafe"
            # call under the default GLOBAL capture mode, which then kills the process ("operation not permitted when stream is
            # capturing", seen on the driver box whenever a collective had run shortly before the capture).  Thread-local mode
            # checks the capturing thread only; the launches of the autograd threads are captured through the stream all the same.
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local" if self.collective else "global"):
                self.loss = self._body()
        except Exception as e:            # noqa: BLE001
            if not self.collective:
                raise
            # a collective that this stack cannot capture: the same launches, host-issued (train_step's static route), on every rank
            # alike (the failure is a property of the software, not of a rank's data)
            import warnings
            self.graph, self.ended, self.capture_error = None, True, f"{type(e).__name__}: {str(e)[:300]}"
            torch.cuda.synchronize()
            for ov in D._SEQ:             # (handles of a dead capture are not waited for)
                ov.work, ov.left, ov.ready = None, len(ov.params), False
            opt.zero_grad(set_to_none=True)
            warnings.warn(f"mirror_nerf_amd: capturing the training step with its collectives failed ({self.capture_error}); "
                          "continuing on train_step's static route", RuntimeWarning, stacklevel=3)
            return
        self._stage_captured = self._stage()
        self._hosts = [torch.zeros(len(self.system.models), dtype=torch.int32).pin_memory() for _ in range(3)]

    def __call__(self, rays, target, gt_mask):
        if not self.ended:
            self._settle(block_beyond=1)
        if not self.ended and self.graph is not None and self._stage_captured != self._stage():
            self.graph = None             # train.py:387-391: the geometry stage ended -- other launches, another graph
        if not self.ended and self.graph is None:
            self._settle(block_beyond=0)  # (nothing of the old graph in flight when the new one is captured)
            if not self.ended:
                self.capture()
        if self.ended:                    # after a range-guard trip (models pinned to fp32) or a failed capture: train_step
            return train_step(self.system, self.opt, rays, target, gt_mask, self.loss_fn, self.epoch, gt_valid=self.gt_valid)
        self.rays.copy_(rays)
        self.target.copy_(target)
        self.gt.copy_(gt_mask)
        self.opt.sync_hyper()
        self.graph.replay()
        self.opt._calls += 1
        for i in self.opt._live_dev:
            self.opt._steps[i] += 1
        from .weights import bump_generation
        bump_generation()                 # host-side caches of the packed images are stale (the replay updated the weights)
        host = self._hosts[self._replays % 3]
        self._replays += 1
        for i, w in enumerate(self._word_views):               # this step's guard words, read one or two calls later
            host[i:i + 1].copy_(w, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, host, [m.__dict__.get("_mnrf_seed_reduction", 0) for m in self.system.models.values()]))
        return self.loss

    def _settle(self, block_beyond=1):
        """Guard words of the replays that have finished (their updates were skipped on the device where a word is set).  The host
        waits only for replays more than `block_beyond` behind the one it is about to queue: with 1 it queues replay N+1 while
        replay N runs and never drains the device (a synchronize() on the previous replay at every call left a launch
        bubble per step)."""
        from .mirror_nerf import GUARD, GUARD_NAMES, _lower_gradient_scale, pin_fp32
        while self._pending:
            ev, host, issued_with = self._pending[0]
            if len(self._pending) > block_beyond:
                ev.synchronize()
            elif not ev.query():
                return
            self._pending.popleft()
            if not GUARD:
                continue
            words = host.tolist()
            models = list(self.system.models.values())
            # (a backward overflow of a replay queued before an adaptation took effect says nothing about the new scale)
            words = [0 if (w and (w & 256) and not (w & (2 | 4 | 128 | 512)) and m.__dict__.get("_mnrf_seed_reduction", 0) > r0) else w
                     for m, w, r0 in zip(models, words, issued_with)]
            if not any(words):
                continue
            import warnings
            if all((not w) or _lower_gradient_scale(m, w) for m, w in zip(models, words)):
                self.graph = None         # scaled activation gradients outgrew f16: the models stay on the split arithmetic with a
                continue                  # lower gradient scale (a launch flag): capture again
            why = "; ".join(v for k, v in GUARD_NAMES.items() if any(w & k for w in words))
            warnings.warn(f"mirror_nerf_amd: the split-f16 arithmetic left its range in a captured training step ({why}); its update was "
                          "skipped, the models run on the fp32 kernels through train_step from now on", RuntimeWarning, stacklevel=3)
            pin_fp32(self.system)
            self.ended = True
            self._pending.clear()
            return


def synthetic_train_bench(dev, all_rays, steps=10, warmup=3, batch=1024, seed=0, loss_name="color_mask", **hp_over):
    """Returns dict(rays_per_s, ms_per_step, reflected_per_step, loss) for this process group.  hp_over: hparams other than
    default_hparams() (e.g. N_importance=128: BASELINE config 3 as worded)."""
    rank, world = D.world()
    route_arg = hp_over.pop("_route", None)
    torch.manual_seed(seed)
    system = NeRFSystem(default_hparams(**hp_over)).to(dev)
    with torch.no_grad():   # opaque density so that surfaces and reflections exist
        for m in (system.nerf_coarse, system.nerf_fine):
            m.sigma.weight.mul_(20.0)
            m.sigma.bias.fill_(1.0)
    import os
    # fused Adam: one kernel per step instead of the foreach kernels, same arithmetic; measured 8.0 vs 8.9 ms per step on the
    # same box (the kernels of a step take 7.9 ms; this removes launch bubbles).  MNRF_ADAM_FUSED=0 selects the foreach implementation.
    # MNRF_FLAT_ADAM=0: torch's fused Adam over the 64 parameter tensors (rounds 2-3); default: the same over one flat tensor per model
    flat_adam = os.environ.get("MNRF_FLAT_ADAM", "1") != "0"
    if flat_adam:
        opt = FlatAdam(list(system.models.values()), lr=5e-4)
    else:
        opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=os.environ.get("MNRF_ADAM_FUSED", "1") == "1")
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    # "run_sh": run.sh:259-280's recipe after its geometry stage -- TotalLoss with all five terms (--use_plane_consistent_loss; epoch
    # >= 4: reflections traced); "run_sh_stage": the same recipe inside the stage (--train_geometry_stage, epochs 0-3; here epoch 2,
    # where all five terms are on: losses.py:233-249): no reflections, target black inside the mirror, stage flag in the loss
    stage = loss_name == "run_sh_stage"
    if stage:
        system.train_geometry_stage = True
    if loss_name == "total":
        loss_fn = total_loss_fn()
    elif loss_name in ("run_sh", "run_sh_stage"):
        loss_fn = total_loss_fn(SimpleNamespace(use_plane_consistent_loss=True), epoch=2 if stage else 5, train_geometry_stage=stage)
    else:
        loss_fn = color_mask_loss
    second_order = loss_name != "color_mask"
    D.attach_overlap(system.models.values())        # (no-op on one rank)

    # route of the step (round 5): "graph" = the whole step replayed as one hipGraph (GraphedTrainStep; one rank), "static" = the
    # same launches issued by the host with no device->host read (train_step(gt_valid=True)), "host" = the reference's shape: the
    # reflected-ray count is read by the host in the middle of the step (train.py:175).  MNRF_TRAIN_ROUTE overrides.
    # With more than one rank the graph holds the collectives too (round 6); a stack that cannot capture them falls back to "static".
    route = os.environ.get("MNRF_TRAIN_ROUTE") or route_arg or "graph"
    if route != "host" and not (flat_adam and static_step_ok(system)):
        route = "host"
    if route == "graph" and (world > 1 or D.forced()) and torch.distributed.get_backend() != "nccl":
        route = "static"      # (gloo stages through the host: nothing to capture; the CPU / shared-GPU test transports)
    graphed = GraphedTrainStep(system, opt, batch, loss_fn, gt_valid=True) if route == "graph" else None

    # The batches are drawn BEFORE the timed region and wait in HBM (bench contract: inputs resident when timing starts; a data
    # loader's prefetch queue): the same draws in the same order as rounds 1-4, which made them inside the loop -- eight small
    # launches per step that were the harness's, not the step's
    batches = []
    for _ in range(warmup + steps):
        idx = torch.randint(0, all_rays.shape[0], (batch,), device=dev, generator=g)
        rays_b = all_rays[idx].contiguous()
        target_b = torch.rand(batch, 3, device=dev, generator=g)
        gt_b = (torch.rand(batch, device=dev, generator=g) < 0.25).float()
        batches.append((rays_b, target_b, gt_b))
    refl_timed = torch.stack([b[2].sum() for b in batches[warmup:]]).sum() * (0.0 if stage else 1.0)      # (no reflections in the stage)
    it = iter(batches)

    def one():
        rays, target, gt = next(it)
        if graphed is not None:
            return graphed(rays, target, gt)
        # (no host read here: it would drain the queue)
        return train_step(system, opt, rays, target, gt, loss_fn, gt_valid=True if route == "static" else None)

    for _ in range(warmup):
        one()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    refl = refl_timed
    for _ in range(steps):
        loss = one()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    refl = float(refl.item())
    # algorithmic work of a step (FLOP = 2 MAC; per sample, every sample of a training render is a full
    # 4-head evaluation): forward 1 318 912 + density-gradient normal 982 528 (compute_normal = trace_secondary_rays,
    # train.py:143) + activation gradients 1 318 912 + weight gradients 1 318 912; with a loss on normal_* keys the
    # second-order pass adds a tangent pass, the density-gradient chain once more and the trunk weight gradients
    # (3 x 982 528).  Priced against the dense f16 MFMA peak (the split path; the dW GEMMs run bf16 x 6 on the same pipe).
    from .mirror_nerf import FLOP_FULL, FLOP_GRAD
    hp = system.hparams
    # field evaluations per ray of a training render: N_samples coarse + (N_samples + N_importance) fine, all full 4-head
    # evaluations (rendering.py:304-360).  run.sh:266 trains with --N_importance 64 (default_hparams): 64 + 128 = 192
    spr = hp.N_samples + (hp.N_samples + hp.N_importance if hp.N_importance > 0 else 0)
    samples = (batch * steps + refl) * spr
    flop_sample = 3 * FLOP_FULL + FLOP_GRAD + (3 * FLOP_GRAD if second_order else 0)
    achieved = samples * flop_sample / dt / 1e12
    roofline = {"bound": "mfma", "achieved": achieved, "peak": 2516.6, "unit": "TFLOP/s", "frac": achieved / 2516.6,
                "flop_per_sample": flop_sample, "samples_per_step": samples / steps,
                "note": "ALGORITHMIC fp32 FLOPs of forward + density-gradient normal + activation gradients + weight gradients"
                        + (" + second-order pass" if second_order else "") + " over the whole step time (host, losses, "
                        "optimizer and all-reduce included); dense f16 MFMA peak"}
    models = list(system.models.values())
    allreduce = None
    n_overlapped = sum(m.__dict__.get("_mnrf_overlap") is not None for m in models)
    if world > 1 or D.forced():
        allreduce = {"buckets_in_place": sum(D._flat_bucket(m) is not None for m in models), "models": len(models),
                     "overlapped_with_backward": n_overlapped,
                     "note": "one flat gradient buffer per model, all-reduced in place (no cat / copy); issued from inside the "
                             "backward pass when a model's last gradient arrives"}
    D.detach_overlap(models)          # (this system is done: its buckets leave the fixed send order)
    return {"value": (batch * steps + refl) * world / dt, "unit": "rays/s (primary+reflected, fwd+bwd+all-reduce+Adam)", "allreduce": allreduce,
            "roofline": roofline, "allreduce_bytes_per_step": 4 * sum(q.numel() for q in params_of(system)) if world > 1 or D.forced() else 0,
            "ms_per_step": dt / steps * 1e3, "batch_rays_per_gpu": batch, "reflected_rays_per_step": refl / steps,
            "samples_per_ray": spr, "N_samples": hp.N_samples, "N_importance": hp.N_importance, "steps": steps,
            "route": (route if not (graphed is not None and graphed.ended) else
                      ("static (graph capture with collectives failed: " + graphed.capture_error + ")" if graphed.capture_error
                       else route + " (ended by a range-guard trip)")),
            "collectives_in_graph": bool(graphed is not None and graphed.collective and not graphed.ended),
            "optimizer": ("training.FlatAdam (Adam over one flat parameter tensor per model, " +
                          ("mnrf_adam_step" if getattr(opt, "kernel", False) else "torch's fused kernel") + ")") if flat_adam else "torch.optim.Adam(fused=True)",
            "loss": float(loss.item()),
            "loss_fn": "losses.TotalLoss (colour, mask, normal, normal_reg; fused HIP kernels; second-order pass on)" if loss_name == "total"
                       else "losses.TotalLoss as run.sh:259-280 trains (colour, mask, plane-consistent with device-side draws, normal, normal_reg)"
                            + (" inside the geometry stage (epoch 2: no reflections, target black inside the mirror)" if stage else " after the geometry stage (epoch 5)")
                       if loss_name in ("run_sh", "run_sh_stage")
                       else "ColorLoss + MirrorMaskLoss of the reference (both typs; fused loss kernel since round 4; no normal_* key read: the second-order pass is skipped)"}
