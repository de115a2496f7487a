#!/usr/bin/env python3
"""A/B of the dynamic tile queue of the 48-samples-per-wave field kernels (FieldArgs::tile_queue; MNRF_TILE_QUEUE=0 keeps the
static one-workgroup-per-tile grid): child processes alternate between the two settings on ONE box; each renders one
32768-ray chunk of the bench workload back to back for `--seconds`, with the 200 ms telemetry running, and reports the mean
HIP-event time of the full (fine-pass) and sigma-only (coarse-pass) launches, the median shader clock, the slowest XCD's clock
and the board power.  What the queue is for: a static grid deals an eighth of the tiles to every XCD and ends when the slowest
XCD is done; with the queue a faster XCD takes more tiles.
    python scripts/ab_tile_queue.py [--rounds 3] [--seconds 6] [--fused]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(seconds, fused):
    import torch
    import bench
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd.telemetry import SmiSampler
    from oracle import mirror_nerf_oracle as O
    dev = torch.device("cuda", 0)
    models, sds, emb = bench.build_models(dev)
    rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + 32768]).to(dev)

    def chunk():
        with torch.no_grad():
            M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False, _maps_only=fused)
    for _ in range(3):
        chunk()
    torch.cuda.synchronize()
    MN.LAUNCH_LOG = []
    t0 = time.perf_counter()
    n = 0
    with SmiSampler(0, 0.2) as smi:
        while time.perf_counter() - t0 < seconds:
            for _ in range(5):
                chunk()
            torch.cuda.synchronize()
            n += 5
    wall = time.perf_counter() - t0
    full = [e0.elapsed_time(e1) for flags, B, e0, e1 in MN.LAUNCH_LOG if not flags & 1]
    sig = [e0.elapsed_time(e1) for flags, B, e0, e1 in MN.LAUNCH_LOG if flags & 1]
    t = smi.summary()
    print("AB " + json.dumps({"queue": os.environ.get("MNRF_TILE_QUEUE", "1"), "chunks": n, "ms_per_chunk": wall / n * 1e3,
                              "full_ms": sum(full) / max(1, len(full)), "sigma_ms": sum(sig) / max(1, len(sig)),
                              "sclk_median": (t.get("sclk_mhz") or {}).get("median"),
                              "sclk_slowest_xcd": (t.get("sclk_slowest_xcd_mhz") or {}).get("median"),
                              "power_w": (t.get("power_w") or {}).get("median")}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--libs", default="", help="comma list of libraries (MNRF_LIB; 'default' = in-tree) to alternate INSTEAD of the queue setting")
    a = ap.parse_args()
    if a.child:
        child(a.seconds, a.fused)
        sys.exit(0)
    variants = [("lib", l) for l in a.libs.split(",")] if a.libs else [("queue", "0"), ("queue", "1")]
    for r in range(a.rounds):
        for kind, q in variants:
            env = dict(os.environ)
            if kind == "queue":
                env["MNRF_TILE_QUEUE"] = q
            elif q != "default":
                env["MNRF_LIB"] = os.path.join(ROOT, q)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--seconds", str(a.seconds)]
                                 + (["--fused"] if a.fused else []), env=env, capture_output=True, text=True, cwd=ROOT)
            lines = [l for l in out.stdout.splitlines() if l.startswith("AB ")]
            print((f"{q:28s} " if kind == "lib" else "") + (lines[-1] if lines else f"{kind}={q}: child failed\n{out.stdout[-1500:]}{out.stderr[-1500:]}"), flush=True)
