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


def _flat_bucket(module):
    """The flat gradient buffer of `module` (autograd._Pending: 32 field parameters in state_dict order) IF every parameter's
    .grad is still the view of it that FieldFn handed to autograd -- then the all-reduce runs on that one tensor in place and
    nobody copies.  None when autograd had to copy (another consumer of a parameter, retain_graph, ...)."""
    flat = module.__dict__.get("_mnrf_flat_grad")
    if flat is None:
        return None
    from .weights import PARAM_NAMES, PARAM_SHAPES, param_refs
    off, offs = 0, {}
    for n in PARAM_NAMES:
        offs[n] = off
        off += int(torch.Size(PARAM_SHAPES[n]).numel())
    base, item = flat.data_ptr(), flat.element_size()
    for sub, pname, full in param_refs(module):
        q = sub._parameters[pname]
        if q is None or not q.requires_grad:
            continue
        g = q.grad
        if g is None or full not in offs or not g.is_contiguous() or g.data_ptr() != base + offs[full] * item:
            return None
    return flat


class _Overlap:
    """All-reduce of a module's bucket issued from INSIDE the backward pass, as soon as the last of its parameters has
    received its gradient (post-accumulate-grad hooks, what DDP's reducer does, train.py:577-584): the fine model's 2.65 MB
    travel over xGMI while the coarse model's backward kernels still run."""

    def __init__(self, module, average):
        from .weights import params_of
        self.module, self.average = module, average
        self.params = [q for q in params_of(module) if q.requires_grad]
        self.left = len(self.params)
        self.work = None
        self.handles = [q.register_post_accumulate_grad_hook(self._hook) for q in self.params]

    def _hook(self, _param):
        self.left -= 1
        if self.left == 0:
            self.left = len(self.params)
            flat = _flat_bucket(self.module)
            if flat is not None and dist.is_initialized():
                self.work = (flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """-> True when this module's gradients have been reduced by the overlapped all-reduce."""
        w, self.work = self.work, None
        self.left = len(self.params)
        if w is None:
            return False
        flat, work = w
        work.wait()
        if self.average:
            flat /= dist.get_world_size()
        return True

    def remove(self):
        for h in self.handles:
            h.remove()


def attach_overlap(modules, average=True):
    """Install the overlapped bucket all-reduce on field modules (MirrorNeRF).  Idempotent; no-op without a process group."""
    out = []
    for m in modules:
        ov = m.__dict__.get("_mnrf_overlap")
        if ov is None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced()):
            ov = _Overlap(m, average)
            m.__dict__["_mnrf_overlap"] = ov
        if ov is not None:
            out.append(ov)
    return out


def allreduce_gradients(params, average=True, modules=()):
    """Sum (average) the gradients over the ranks.  Field modules passed in `modules` are reduced through their flat
    gradient buffer -- ONE tensor per module whose views are the .grads (autograd._Pending): no torch.cat, no copy back; the
    all-reduce of a module with attach_overlap() was already issued during the backward pass and is only waited for here.
    Whatever is left (parameters outside those modules, or a module whose .grads autograd had to copy) goes the generic
    way: one flat copy, one all-reduce, copied back.  RCCL over xGMI: direct reduce-scatter + all-gather inside RCCL uses
    all 7 links; a 2.65 MB message is latency-, not bandwidth-bound."""
    rank, ws = world()
    if ws == 1 and not forced():
        return
    done = set()
    pending = []
    for m in modules:
        ov = m.__dict__.get("_mnrf_overlap")
        reduced = ov.finish() if ov is not None else False
        flat = _flat_bucket(m)
        if flat is None:
            continue
        if not reduced:
            pending.append((flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
        from .weights import params_of
        done.update(id(q) for q in params_of(m))
    for flat, work in pending:
        work.wait()
        if average:
            flat /= ws
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
