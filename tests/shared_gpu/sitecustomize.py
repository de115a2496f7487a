"""TEST AID, not product code.  A directory put on PYTHONPATH by tests/test_dist_gpu.py (and by hand: PYTHONPATH=tests/shared_gpu
MNRF_SHARE_GPU=1 python bench.py --gpus 2): Python imports `sitecustomize` at start-up, and with MNRF_SHARE_GPU=1 this one makes
N ranks share GPU 0 of a 1-GPU box over gloo, so that the N > 1 code paths of mirror_nerf_amd.dist / training / bench.py -- ray
sharding, per-rank batches, the bucket all-reduce issued from the backward hooks, the guard-flag collective, max-over-ranks
timing -- execute on device tensors.  RCCL refuses two ranks on one device; TIMINGS UNDER THIS AID MEAN NOTHING.

Round 4 kept these branches inside mirror_nerf_amd/dist.py and training.py; they are a transport shim and live here now:
  * dist.init_from_env: every rank binds GPU 0, the group is "gloo";
  * every BLOCKING torch.distributed.all_reduce drains the device first.  Four processes time-slicing one GPU stall inside gloo
    when a blocking collective is issued behind asynchronous ones still in flight (scripts/repro_gloo_shared_gpu.py reproduces it
    with no kernel of this package; profiles/r05_gloo_shared_gpu.txt) -- a property of this aid's transport, which RCCL (one
    process per GPU, stream-ordered, no host staging) does not share.
"""
import os
import sys

if os.environ.get("MNRF_SHARE_GPU") == "1":
    _root = os.environ.get("MNRF_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if _root not in sys.path:
        sys.path.insert(0, _root)
    import torch
    import torch.distributed as dist
    from mirror_nerf_amd import dist as D

    def _init_from_env(device=None):
        ws = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        if (ws > 1 or os.environ.get("MNRF_FORCE_COLLECTIVES", "0") == "1") and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=rank, world_size=ws)
        return rank, ws, device

    _all_reduce = dist.all_reduce

    def _drained_all_reduce(tensor, *a, **k):
        if not k.get("async_op", False) and tensor.is_cuda:
            torch.cuda.synchronize()
        return _all_reduce(tensor, *a, **k)

    D.init_from_env = _init_from_env
    dist.all_reduce = _drained_all_reduce
    torch.distributed.all_reduce = _drained_all_reduce
