#!/usr/bin/env python3
"""Minimal reproducer for the stall of round 4's shared-GPU test aid (profiles/r04zk_README.txt): N processes time-slicing ONE
GPU, torch.distributed over gloo, per iteration two ASYNC all-reduces of 2.65 MB device tensors in flight plus one BLOCKING
all-reduce of a small device tensor behind them -- the collective pattern of training.train_step (two gradient buckets from the
backward hooks, then the range-guard flags).  NOTHING of mirror_nerf_amd is imported: the only GPU work is torch.matmul.

    python scripts/repro_gloo_shared_gpu.py --ranks 4 [--drain] [--iters 200]

Exit 0: every rank finished.  Exit 3: a rank sat in one iteration for more than --stall seconds (its stack is dumped).
--drain: torch.cuda.synchronize() before the blocking collective (what tests/shared_gpu/sitecustomize.py does)."""
import argparse
import faulthandler
import os
import socket
import sys
import threading
import time


def worker(rank, ws, port, a, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    dev = torch.device("cuda", 0)
    x = torch.randn(2048, 2048, device=dev)
    b1 = torch.ones(662152, device=dev)
    b2 = torch.ones(662152, device=dev)
    state = {"it": -1, "t": time.time()}

    def watchdog():
        while True:
            time.sleep(2.0)
            if time.time() - state["t"] > a.stall:
                sys.stderr.write(f"[rank {rank}] STALL in iteration {state['it']} for more than {a.stall} s\n")
                faulthandler.dump_traceback(file=sys.stderr)
                q.put((rank, "stall", state["it"]))
                os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    for it in range(a.iters):
        state["it"], state["t"] = it, time.time()
        y = x
        for _ in range(a.matmuls):
            y = torch.matmul(y, x) * 1e-3          # queued GPU work the collectives sit behind
        w1 = dist.all_reduce(b1, async_op=True)
        w2 = dist.all_reduce(b2, async_op=True)
        flag = (y[0, :3] != 12345.0).float()
        if a.drain:
            torch.cuda.synchronize()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)      # blocking, behind two async ones still in flight
        w1.wait()
        w2.wait()
        b1.fill_(1.0)
        b2.fill_(1.0)
    torch.cuda.synchronize()
    dist.barrier()
    q.put((rank, "done", a.iters))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--matmuls", type=int, default=4)
    ap.add_argument("--stall", type=float, default=45.0)
    ap.add_argument("--drain", action="store_true")
    a = ap.parse_args()
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=worker, args=(r, a.ranks, port, a, q)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    results = []
    deadline = time.time() + a.stall * 3 + a.iters * 1.0
    while len(results) < a.ranks and time.time() < deadline:
        try:
            results.append(q.get(timeout=5))
        except Exception:  # noqa: BLE001
            if any(r[1] == "stall" for r in results):
                break
    stalled = [r for r in results if r[1] == "stall"]
    for p in procs:
        p.join(2)
        if p.is_alive():
            p.kill()      # (the exact processes this script started)
    verdict = "STALL" if stalled or len(results) < a.ranks else "ok"
    print(f"repro_gloo_shared_gpu ranks={a.ranks} drain={a.drain} iters={a.iters}: {verdict} "
          f"({sorted(results)}) in {time.time() - t0:.1f} s", flush=True)
    sys.exit(3 if verdict == "STALL" else 0)


if __name__ == "__main__":
    main()
