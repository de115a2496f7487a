"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests).

Rendering shards naturally -- rays are independent units -- so the data path has NO collective:
each rank renders an interleaved set of ray tiles (mirror pixels cluster spatially; interleaving
balances the reflected-ray load) and owns the matching slice of every output map.  Collectives
appear only (a) optionally, to assemble a frame on rank 0, and (b) in training, as ONE all-reduce
of the flat gradient buffer (1 324 304 fp32 = 5.3 MB for the coarse+fine pair; the reference
gets the same from Lightning DDP, train.py:577-584).
"""
import os

# The host driver of the MI355X boxes only supports dmabuf IPC: RCCL (and any CUDA-tensor sharing between processes) fails
# with `hipIpcGetMemHandle: invalid argument` unless this is set BEFORE the HIP/HSA runtime initialises -- i.e. before
# the first torch.cuda call of the process, which init_from_env() itself makes.  Hence at import time, not there.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TILE = 4096


def forced():
    """MNRF_FORCE_COLLECTIVES=1: issue the collectives even in a 1-rank group (lets a single-GPU box exercise the
    RCCL code path that the 8-GPU run uses)."""
    return os.environ.get("MNRF_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized()


def init_from_env(device=None):
    """One process per GPU (train.py:577-584 gets the same from Lightning DDP): reads RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as set by torch.distributed.run, binds this process to GPU LOCAL_RANK and brings up the RCCL group
    (backend "nccl" IS RCCL on ROCm).  A 1-rank group is only created when MNRF_FORCE_COLLECTIVES=1.
    (The round-4 test aid that put several ranks on one GPU over gloo lives in tests/shared_gpu/ now: it replaces this
    function from a sitecustomize hook of the test processes; nothing in the package knows about it.)
    Returns (rank, world_size, device)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    want_group = ws > 1 or (os.environ.get("MNRF_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ)
    if want_group and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if device.type == "cuda":
            dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=ws)
    return rank, ws, device


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_rays, rank, world_size, tile=TILE, device="cpu"):
    """Ray indices of `rank`: tiles of `tile` consecutive rays dealt round-robin to the ranks."""
    idx = torch.arange(n_rays, device=device)
    return idx[(idx // tile) % world_size == rank]


def render_sharded(render_fn, rays, tile=TILE):
    """Render this rank's tiles.  render_fn(rays_subset) -> dict of per-ray tensors.
    Returns (indices, results) -- results cover rays[indices] only; no communication."""
    rank, ws = world()
    idx = shard_indices(rays.shape[0], rank, ws, tile, rays.device)
    return idx, render_fn(rays[idx].contiguous())


def shard_count(n_rays, rank, world_size, tile=TILE):
    """len(shard_indices(...)) without building the index."""
    full, rem = divmod(n_rays, tile)
    n = (full // world_size + (1 if rank < full % world_size else 0)) * tile
    return n + (rem if full % world_size == rank else 0)


def gather_frame(idx, results, n_rays, keys=None, dst=0, tile=TILE):
    """Optional assembly of full per-ray maps on rank `dst` (20 B/ray for rgb+depth+mask at most).  `tile` must be the
    tile size `idx` was dealt with (render_sharded's)."""
    rank, ws = world()
    keys = list(results) if keys is None else keys
    if ws == 1 and not forced():
        return {k: results[k] for k in keys}
    out = {}
    counts = [shard_count(n_rays, r, ws, tile) for r in range(ws)]
    if counts[rank] != idx.numel():
        raise RuntimeError(f"gather_frame: this rank holds {idx.numel()} rays but tile={tile} deals it {counts[rank]}")
    idx_parts = [torch.empty(c, dtype=idx.dtype, device=idx.device) for c in counts]
    dist.all_gather(idx_parts, idx) if len(set(counts)) == 1 else _all_gather_ragged(idx_parts, idx, counts)
    for k in keys:
        v = results[k]
        parts = [torch.empty((c,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for c in counts]
        dist.all_gather(parts, v.contiguous()) if len(set(counts)) == 1 else _all_gather_ragged(parts, v.contiguous(), counts)
        if rank == dst:
            full = torch.empty((n_rays,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            for ip, p in zip(idx_parts, parts):
                full[ip] = p
            out[k] = full
    return out


def _all_gather_ragged(parts, mine, counts):
    """all_gather for unequal shard sizes: broadcast each rank's piece in turn."""
    rank, ws = world()
    for r in range(ws):
        if r == rank:
            parts[r].copy_(mine)
        dist.broadcast(parts[r], src=r)


def max_over_ranks(seconds, device):
    """The slowest rank's time (bench.py contract)."""
    rank, ws = world()
    if ws == 1 and not forced():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _field_layout(module):
    """{qualified parameter name: (offset, numel)} of the flat gradient buffer of a FIELD module (autograd._Pending: the 32
    parameters of weights.PARAM_NAMES in state_dict order, absent optional heads included as zero slots), total size --
    or (None, 0) when `module` is something else (e.g. the hash-grid model)."""
    from .weights import PARAM_NAMES, PARAM_SHAPES, param_refs
    names = {full for _, _, full in param_refs(module)}
    if not names or not names <= set(PARAM_NAMES):
        return None, 0
    off, lay = 0, {}
    for n in PARAM_NAMES:
        k = int(torch.Size(PARAM_SHAPES[n]).numel())
        lay[n] = (off, k)
        off += k
    return lay, off


def _flat_bucket(module):
    """The flat gradient buffer of `module` (autograd._Pending) IF every parameter's .grad is still the view of it that
    FieldFn handed to autograd -- then the all-reduce runs on that one tensor in place and nobody copies.  None when
    autograd had to copy (another consumer of a parameter, retain_graph, ...) or a parameter has no gradient."""
    flat = module.__dict__.get("_mnrf_flat_grad")
    if flat is None:
        return None
    lay, total = _field_layout(module)
    if lay is None or flat.numel() != total:
        return None
    from .weights import param_refs
    base, item = flat.data_ptr(), flat.element_size()
    for sub, pname, full in param_refs(module):
        q = sub._parameters[pname]
        if q is None or not q.requires_grad:
            continue
        g = q.grad
        if g is None or not g.is_contiguous() or g.data_ptr() != base + lay[full][0] * item:
            return None
    return flat


def _is_hashgrid(module):
    from .mirror_nerf_tcnn import MirrorNeRFTcnn
    return isinstance(module, MirrorNeRFTcnn)


def _module_message(module):
    """-> (flat tensor to all-reduce, copied?).  EVERY rank sends one message of the same size and element order per module
    and step, whatever happened to its .grads locally: the in-place bucket when they still alias it, else a flat copy in
    the same layout (zeros where this rank has no gradient) -- ranks may take different branches, the collective matches."""
    flat = _flat_bucket(module)
    if flat is not None:
        return flat, False
    from .weights import param_refs, params_of
    lay, total = _field_layout(module)
    ps = [q for q in params_of(module) if q is not None and q.requires_grad]
    ref = next((q.grad for q in ps if q.grad is not None), ps[0] if ps else None)
    if ref is None:
        return None, False
    if lay is None:      # not a field module: parameters() order
        total = sum(q.numel() for q in ps)
        flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for q in ps:
            if q.grad is not None:
                flat[off:off + q.numel()].copy_(q.grad.reshape(-1))
            off += q.numel()
        return flat, True
    flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
    for sub, pname, full in param_refs(module):
        q = sub._parameters[pname]
        if q is not None and q.requires_grad and q.grad is not None:
            o, k = lay[full][0], q.numel()      # (a model with fewer encoding bands fills a prefix of its slot, on every rank alike)
            flat[o:o + k].copy_(q.grad.reshape(-1))
    return flat, True


def _module_messages(module):
    """-> list of (flat tensor to all-reduce, deliver(flat) or None).  One message for a field module (_module_message).
    TWO for the hash-grid model (BASELINE config 5; reference: DDP over all parameters, train.py:577-584): its table gradient
    -- 2 x 6.1 M floats that already ARE one tensor -- is reduced IN PLACE (no cat, no copy back), and the eleven small MLP
    tensors (11 k floats) travel as one flat blob.  Every rank sends both, zeros where it holds no gradient."""
    if _is_hashgrid(module):
        table = module.encoder.embeddings
        if table.grad is None or not table.grad.is_contiguous():
            table.grad = torch.zeros_like(table) if table.grad is None else table.grad.contiguous()
        mlp = [q for q in module.mlp_params() if q.requires_grad]
        blob = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in mlp])

        def deliver(flat, mlp=mlp):
            off = 0
            for q in mlp:
                v = flat[off:off + q.numel()].view_as(q)
                q.grad = v.clone() if q.grad is None else q.grad.copy_(v)
                off += q.numel()
        return [(table.grad, None), (blob, deliver)]
    flat, copied = _module_message(module)
    if flat is None:
        return []
    return [(flat, (lambda f, m=module: _scatter_message(m, f)) if copied else None)]


def _scatter_message(module, flat):
    """Reduced flat copy -> the module's .grads (a parameter without a local gradient receives the other ranks' mean)."""
    from .weights import param_refs, params_of
    lay, _ = _field_layout(module)
    if lay is None:
        off = 0
        for q in (q for q in params_of(module) if q is not None and q.requires_grad):
            v = flat[off:off + q.numel()].view_as(q)
            q.grad = v.clone() if q.grad is None else q.grad.copy_(v)
            off += q.numel()
        return
    for sub, pname, full in param_refs(module):
        q = sub._parameters[pname]
        if q is not None and q.requires_grad:
            o, k = lay[full][0], q.numel()
            v = flat[o:o + k].view_as(q)
            q.grad = v.clone() if q.grad is None else q.grad.copy_(v)


# The fixed order in which the module buckets go out.  Collectives must be issued in the SAME order on every rank; readiness
# is not the same on every rank (a rank whose batch holds no mirror pixel builds a different graph; a parameter may receive
# no gradient at all; autograd may have copied a .grad on one rank only), so the order is the order of attach_overlap() --
# code, not data -- and a bucket goes out from inside the backward pass only when all buckets before it have gone out.
_SEQ = []


class _Overlap:
    """All-reduce of a module's bucket issued from INSIDE the backward pass, as soon as the last of its parameters has
    received its gradient (post-accumulate-grad hooks, what DDP's reducer does, train.py:577-584): the fine model's 2.65 MB
    travel over xGMI while the coarse model's backward kernels still run."""

    def __init__(self, module, average):
        from .weights import params_of
        self.module, self.average = module, average
        self.params = [q for q in params_of(module) if q.requires_grad]
        self.left = len(self.params)
        self.ready = False
        self.work = None          # [(flat, handle, deliver)] once issued in this step
        self.enabled = True       # False under no_overlap(): hooks only count, allreduce_gradients() sends everything
        self.handles = [q.register_post_accumulate_grad_hook(self._hook) for q in self.params]

    def _hook(self, _param):
        if not self.enabled:
            return
        if self.left <= 0 or self.work is not None:
            # a second backward pass reached this module before allreduce_gradients() collected the first one's bucket
            # (gradient accumulation, a recomputed step): its all-reduce runs -- or ran -- on the very buffer this pass is
            # adding to.  Refuse loudly instead of reducing half-accumulated sums; recomputing callers call reset_overlap().
            raise RuntimeError("mirror_nerf_amd.dist: a second backward pass started before allreduce_gradients() finished the "
                               "overlapped all-reduce of the first; call dist.reset_overlap() to discard it (a recomputed "
                               "step) or run the accumulation passes under dist.no_overlap()")
        self.left -= 1
        if self.left == 0:
            self.ready = True
            if self.enabled:
                _issue_ready_prefix()

    def reset(self):
        """Drain and discard this step's state (a pending all-reduce is waited for so that nobody writes under it)."""
        w, self.work = self.work, None
        for _flat, work, _deliver in (w or ()):
            work.wait()
        self.left, self.ready = len(self.params), False

    def issue(self):
        if self.work is None and dist.is_initialized():
            msgs = _module_messages(self.module)
            if _TRACE:
                _trace(f"bucket {_SEQ.index(self) if self in _SEQ else -1} {type(self.module).__name__} left={self.left} "
                       f"sizes={[int(f.numel()) for f, _ in msgs]}")
            if msgs:
                self.work = [(flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), deliver) for flat, deliver in msgs]

    def finish(self):
        """Wait for this module's all-reduce(s) (issued by now) and deliver them; resets the per-step state."""
        w, self.work = self.work, None
        self.left, self.ready = len(self.params), False
        if w is None:
            return False
        for flat, work, deliver in w:
            work.wait()
            if self.average:
                flat /= dist.get_world_size()
            if deliver is not None:
                deliver(flat)
        return True

    def remove(self):
        for h in self.handles:
            h.remove()
        if self in _SEQ:
            _SEQ.remove(self)


_TRACE = os.environ.get("MNRF_DIST_TRACE", "0") == "1"      # every collective this module (and train_step) issues, per rank, to stderr


def _trace(what):
    import sys
    import threading
    print(f"[mnrf.dist rank {dist.get_rank()} {threading.current_thread().name}] {what}", file=sys.stderr, flush=True)


def issue_pending():
    """Send every bucket the backward pass has not sent from its hooks, in the fixed order.  A caller that issues a collective
    of its own between backward() and allreduce_gradients() (train_step's range-guard flag) calls this FIRST: whether a bucket
    was ready inside the backward pass is local to a rank, so anything slipped in between would sit at different positions of
    the ranks' collective sequences."""
    if dist.is_available() and dist.is_initialized():
        for ov in _SEQ:
            if ov.enabled:
                ov.issue()


def _issue_ready_prefix():
    for ov in _SEQ:
        if ov.work is not None:
            continue
        if not ov.ready:
            break
        ov.issue()


def reset_overlap():
    """Discard the overlapped all-reduces of a backward pass whose gradients will not be used (train_step recomputing a step
    after a range-guard trip): every pending collective is waited for, counters start over.  Collective-safe as long as every
    rank calls it at the same point (train_step makes the recompute decision with an all-reduce)."""
    for ov in _SEQ:
        ov.reset()


def detach_overlap(modules=None):
    """Remove the overlapped all-reduce hooks of `modules` (None: of every module): buckets leave in the fixed order of _SEQ, so
    modules that no longer take part in the step (a finished benchmark leg, a discarded model) must not stay in it -- a bucket
    that never becomes ready holds back the ones behind it until allreduce_gradients()."""
    mods = None if modules is None else {id(m) for m in modules}
    for ov in list(_SEQ):
        if mods is None or id(ov.module) in mods:
            ov.reset()
            ov.remove()
            ov.module.__dict__.pop("_mnrf_overlap", None)


class no_overlap:
    """Context manager for gradient accumulation: backward passes inside it do not send buckets from their hooks (and may
    run more than once); allreduce_gradients() after the LAST pass sends the accumulated buffers."""

    def __enter__(self):
        for ov in _SEQ:
            ov.enabled = False
        return self

    def __exit__(self, *exc):
        for ov in _SEQ:
            ov.enabled = True
            ov.left, ov.ready = len(ov.params), False
        return False


def attach_overlap(modules, average=True):
    """Install the overlapped bucket all-reduce on field modules (MirrorNeRF) or hash-grid models (MirrorNeRFTcnn: the table
    gradient in place + one blob of the MLP gradients).  Idempotent; no-op without a process group.
    Pass the modules in FORWARD order (coarse, fine): their buckets go out in the reverse -- the order in which a backward
    pass completes them -- and that order must be the same on every rank (see _SEQ)."""
    out = []
    fresh = []
    for m in modules:
        ov = m.__dict__.get("_mnrf_overlap")
        if ov is None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced()):
            ov = _Overlap(m, average)
            m.__dict__["_mnrf_overlap"] = ov
            fresh.append(ov)
        if ov is not None:
            out.append(ov)
    _SEQ.extend(reversed(fresh))
    return out


def allreduce_gradients(params, average=True, modules=()):
    """Sum (average) the gradients over the ranks.  Field modules passed in `modules` are reduced through their flat
    gradient buffer -- ONE tensor per module whose views are the .grads (autograd._Pending): no torch.cat, no copy back; the
    all-reduce of a module with attach_overlap() was already issued during the backward pass and is only waited for here.
    Every rank sends exactly one message per module, of one size, in one order (_module_message, _SEQ), so ranks whose
    local state differs still match.  Parameters outside `modules` go the generic way: one flat copy, one all-reduce,
    copied back.  RCCL over xGMI: direct reduce-scatter + all-gather inside RCCL uses all 7 links; a 2.65 MB message is
    latency-, not bandwidth-bound."""
    rank, ws = world()
    if ws == 1 and not forced():
        return
    from .weights import params_of
    done = set()
    with_ov = [m.__dict__["_mnrf_overlap"] for m in modules if m.__dict__.get("_mnrf_overlap") is not None]
    with_ov += [ov for ov in _SEQ if ov.work is not None and ov not in with_ov]      # (sent by its hooks but not listed: still waited for)
    for ov in _SEQ:                      # whatever the backward pass did not send goes out now, in the fixed order
        if ov in with_ov:
            ov.issue()
    plain = []
    for m in modules:                    # modules without hooks: same uniform message(s), issued in the order given
        if m.__dict__.get("_mnrf_overlap") is None:
            for flat, deliver in _module_messages(m):
                plain.append((m, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), deliver))
    for ov in _SEQ:
        if ov in with_ov and ov.finish():
            done.update(id(q) for q in params_of(ov.module))
    for m, flat, work, deliver in plain:
        work.wait()
        if average:
            flat /= ws
        if deliver is not None:
            deliver(flat)
        done.update(id(q) for q in params_of(m))
    grads = [p.grad for p in params if p.grad is not None and id(p) not in done]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= ws
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
