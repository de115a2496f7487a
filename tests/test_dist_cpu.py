"""world_size-2 gloo tests of the multi-GPU plumbing (runs on CPU, no GPU needed)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mirror_nerf_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        torch.manual_seed(0)
        rays = torch.randn(n_rays, 8)

        def fake_render(r):   # any per-ray function: sharding must be transparent
            return {"rgb": r[:, :3] * 2 + 1, "depth": r[:, 6] - r[:, 7]}

        idx, res = D.render_sharded(fake_render, rays, tile=D.TILE)
        full = D.gather_frame(idx, res, n_rays)
        idx2, res2 = D.render_sharded(fake_render, rays, tile=1000)      # a non-default tile must size the gather buffers
        full2 = D.gather_frame(idx2, res2, n_rays, tile=1000)
        t = D.max_over_ranks(1.0 + rank, "cpu")
        p = torch.nn.Parameter(torch.zeros(5))
        p.grad = torch.full((5,), float(rank + 1))
        p2 = torch.nn.Parameter(torch.zeros(2, 3))
        p2.grad = torch.full((2, 3), 10.0 * (rank + 1))
        D.allreduce_gradients([p, p2])
        ok = True
        if rank == 0:
            want = fake_render(rays)
            ok = all(torch.equal(full[k], want[k]) and torch.equal(full2[k], want[k]) for k in want)
        q.put((rank, int(idx.numel()), ok, t, p.grad.tolist(), p2.grad[0, 0].item()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_gather_and_gradient_allreduce():
    n_rays = 3 * D.TILE + 123     # ragged: ranks get different counts
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out[0][1] + out[1][1] == n_rays          # every ray rendered exactly once
    assert out[0][1] == 2 * D.TILE and out[1][1] == D.TILE + 123
    assert out[0][2] is True                         # gathered frame == unsharded render
    assert out[0][3] == 2.0 and out[1][3] == 2.0     # max over ranks
    assert out[0][4] == [1.5] * 5 and out[1][5] == 15.0   # averaged gradients


def test_shard_indices_partition():
    for n, ws in ((10, 1), (4096 * 5 + 7, 2), (4096 * 9, 4), (100, 8)):
        seen = torch.cat([D.shard_indices(n, r, ws) for r in range(ws)])
        assert torch.equal(torch.sort(seen)[0], torch.arange(n))


def test_shard_count_matches_indices():
    for n in (0, 1, 4095, 4096, 4097, 4096 * 5 + 7, 640000):
        for ws in (1, 2, 3, 8):
            for tile in (4096, 1000):
                for r in range(ws):
                    assert D.shard_count(n, r, ws, tile) == D.shard_indices(n, r, ws, tile).numel(), (n, ws, tile, r)
