#!/usr/bin/env python3
"""bench.py -- throughput of the rendering hot path on MI355X.

Workload (BASELINE.json configs[1]): novel-view rendering of the synthetic 800x800 Blender-style
camera with the reference's eval procedure -- eval.batched_inference (eval.py:114-740): 64 coarse
(sigma-only) + 192 fine samples per ray, chunk 32768, one reflection bounce.  Weights are the
random-init 8x256 MirrorNeRF pair (torch.manual_seed(0)) with the density made opaque and the
mirror head biased to 1, so that -- as eval.py does whenever a chunk contains mirror pixels --
every primary ray spawns one reflected ray: 640 000 primary + 640 000 reflected rays per frame.
A "step" is one frame.  Inputs (rays, packed weights) are resident in HBM when timing starts; the
result maps stay on the GPU (`to_cpu=False`; the reference's per-chunk D2H of every dict entry
is caller-side data movement).

Multi-GPU (`--gpus N`, launched by torch.distributed.run): frames are independent units; every
rank renders its own frame (weak scaling), no collective on the data path, barrier + max-over-ranks
timing.  value = rays of all ranks / slowest rank's time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
N_SAMPLES, N_IMPORTANCE, CHUNK = 64, 128, 32768
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # same guide, dense f16/bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
# MFMA FLOPs the split kernel EXECUTES per full sample: 1308 hi/lo tile pairs x 3 products x 2 groups x 4 waves x
# 16384 FLOP per v_mfma_f32_16x16x32_f16 / 128 samples (= 3 x the padded fp32 count; algorithmic: MN.FLOP_FULL)
SPLIT_EXECUTED_FLOP_FULL = 1308 * 3 * 2 * 4 * 16384 // 128
# HBM bytes per full-kernel sample from the PMC passes (profiles/r01_pmc for the fp32 kernel, profiles/r01e_pmc_split
# for the split kernel: the same traffic; FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE):
# (2 x 33758 + 196608) KiB / 6291456 samples
PMC_HBM_BYTES_PER_FULL_SAMPLE = (2 * 33758 + 196608) * 1024 / 6291456


def build_models(dev):
    import torch
    import mirror_nerf_amd as M
    from tests.golden import weights as GW
    sds = [GW.apply_tweaks(sd, GW.ALL_MIRROR) for sd in GW.make_state_dict(0, 2)]
    models = {}
    for name, sd in zip(("coarse", "fine"), sds):
        m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        models[name] = m.to(dev)
    return models, sds, {"xyz": M.Embedding(10), "dir": M.Embedding(4)}


ARGS = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)


def cpu_baseline(sds, n_rays=1024):
    """The oracle (numpy port of the reference path) on a bounded sample of the same workload:
    `n_rays` primary rays of the frame + their reflected rays.  The fp32 GEMM backend is whichever
    of numpy/OpenBLAS and torch's CPU sgemm (what the reference's CPU path runs on) is faster on
    this host in a short trial; threads = min(host cores, 64)."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import mirror_nerf_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    rays = O.synthetic_rays(H, W)
    idx = np.linspace(0, rays.shape[0] - 1, n_rays).astype(np.int64)
    sub = rays[idx]
    models = {"coarse": sds[0], "fine": sds[1]}

    def torch_sgemm(x, w):
        return torch.mm(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(w).t()).numpy()

    def run(r):
        t0 = time.perf_counter()
        out = O.render_eval(models, {"xyz": 10, "dir": 4}, r, N_SAMPLES, N_IMPORTANCE, False, CHUNK, ARGS)
        return time.perf_counter() - t0, out

    with threadpool_limits(limits=threads):
        trial = {}
        for name, fn in (("numpy-openblas", None), ("torch-cpu-sgemm", torch_sgemm)):
            O.set_sgemm(fn)
            run(sub[:32])                        # warm-up
            trial[name] = run(sub[:128])[0]
        best = min(trial, key=trial.get)
        O.set_sgemm(torch_sgemm if best == "torch-cpu-sgemm" else None)
        dt, r = run(sub)
        O.set_sgemm(None)
    traced = int((r["mirror_mask_fine"] != 0).any()) * n_rays
    return {"value": (n_rays + traced) / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{n_rays} primary + {traced} reflected rays of the same frame, oracle with {best} "
                      f"({threads} threads), {dt:.1f} s"}


def hash_grid_leg(dev, rays):
    """BASELINE config 5 for the record (not `value`): the hash-grid field (MirrorNeRFTcnn, bound 6, 2^19 x 16 x 2 table)
    on the same 800x800 rays -- one frame of primary rays through render_rays (64 sigma-only + 192 full samples), and the
    1024-ray training step (forward + hand-written backward + Adam)."""
    import torch
    import mirror_nerf_amd as M
    torch.manual_seed(0)
    models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev)
              for k in ("coarse", "fine")}
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}

    def frame():
        with torch.no_grad():
            for c in range(0, rays.shape[0], CHUNK):
                M.render_rays(models, emb, rays[c:c + CHUNK], N_SAMPLES, False, 0, 0, N_IMPORTANCE, CHUNK, test_time=True,
                              compute_normal=False)
    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame()
    torch.cuda.synchronize()
    dt_f = time.perf_counter() - t0
    opt = torch.optim.Adam([p for m in models.values() for p in m.parameters()], lr=5e-4)
    target = torch.rand(1024, 3, device=dev)

    def step():
        idx = torch.randint(0, rays.shape[0], (1024,), device=dev)
        res = M.render_rays(models, emb, rays[idx], N_SAMPLES, False, 1, 1, N_IMPORTANCE, compute_normal=False)
        loss = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean() \
            + 0.1 * ((res["mirror_mask_fine"] - 0.5) ** 2).mean() + 1e-4 * res["surface_normal_fine"].pow(2).sum(-1).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt_t = (time.perf_counter() - t0) / 10
    n = rays.shape[0]
    return {"rays_per_s": n / dt_f, "samples_per_s": n * (2 * N_SAMPLES + N_IMPORTANCE) / dt_f, "frame_ms": dt_f * 1e3,
            "train_ms_per_step": dt_t * 1e3, "train_rays_per_s": 1024 / dt_t,
            "note": "MirrorNeRFTcnn pair, random init; primary rays only (a random-init mask head predicts no mirror); parity "
                    "against tinycudann unpinned (DESIGN.md 4.3); VALU + gather kernels, the MLPs are not on MFMA yet"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096)
    ap.add_argument("--no-train", action="store_true", help="skip the short training-step measurement")
    ap.add_argument("--precision", choices=("split", "fp32"), default="split",
                    help="arithmetic of the field kernel's Linears: fp32 operands as hi/lo f16 pairs on the f16 matrix "
                         "pipe (default, ~1e-6 of fp32) or the bit-exact fp32 MFMA chain")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or (os.environ.get("MNRF_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from oracle import mirror_nerf_oracle as O

    MN.set_precision(a.precision)
    models, sds, emb = build_models(dev)
    # every rank renders its own view: same camera model, pose rotated about z by the rank index
    pose = O.look_at_pose(eye=(4.0 * np.sin(0.3 * rank), -4.0 * np.cos(0.3 * rank), 1.5))
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    rays = torch.empty(H * W, 8, device=dev)
    import ctypes
    c2w = (ctypes.c_float * 12)(*pose.reshape(-1).tolist())
    M._lib.check(M._lib.lib().mnrf_generate_rays(H, W, float(focal), c2w, 0.05, 8.0, M._lib.ptr(rays),
                                                 M._lib.stream()), "mnrf_generate_rays")

    def frame():
        return M.batched_inference(models, emb, rays, N_SAMPLES, N_IMPORTANCE, False, CHUNK, args=ARGS,
                                   trace_secondary_rays=True, to_cpu=False)

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        out = frame()
    sync()
    MN.LAUNCH_LOG = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = frame()
    sync()
    dt = time.perf_counter() - t0
    log, MN.LAUNCH_LOG = MN.LAUNCH_LOG, None
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_mirror = int((out["mirror_mask_fine"] != 0).sum().item())
    out = None
    rays_per_frame = H * W + (H * W if n_mirror > 0 else 0)      # eval.py:159: level 0 traces every ray
    evals_per_ray = N_SAMPLES + (N_SAMPLES + N_IMPORTANCE)
    total_rays = rays_per_frame * a.steps * world
    value = total_rays / dt

    # dominant kernel: the full (4-head) field kernel of the fine pass
    full = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in log if not (flags & 1)]
    sig = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in log if (flags & 1)]
    ms_full = sum(t for _, t in full)
    flop_full = sum(B for B, _ in full) * MN.FLOP_FULL
    achieved = flop_full / (ms_full * 1e-3) / 1e12 if ms_full > 0 else 0.0
    ms_sig = sum(t for _, t in sig)
    sig_tf = sum(B for B, _ in sig) * MN.FLOP_SIGMA / (ms_sig * 1e-3) / 1e12 if ms_sig > 0 else 0.0

    # the other arithmetic on ONE whole frame of the same workload (rank 0), for the record: rays/s and the dominant
    # kernel's rate under it
    other = "fp32" if a.precision == "split" else "split"
    other_tf = other_rays = other_ms = None
    if rank == 0:
        MN.set_precision(other)
        frame()                      # warm-up (first launches of the other kernels)
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        t1 = time.perf_counter()
        frame()
        torch.cuda.synchronize()
        dt_o = time.perf_counter() - t1
        t_o = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in MN.LAUNCH_LOG if not (flags & 1)]
        other_tf = sum(B for B, _ in t_o) * MN.FLOP_FULL / (sum(t for _, t in t_o) * 1e-3) / 1e12
        other_rays = (H * W * 2) / dt_o
        other_ms = sum(t for _, t in t_o) / max(1, len(t_o))
        MN.LAUNCH_LOG = None
        MN.set_precision(a.precision)

    # PCIe-inclusive rate (never `value`): one more frame whose per-ray maps are copied to the host
    host_maps = None
    if rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hm = M.batched_inference(models, emb, rays, N_SAMPLES, N_IMPORTANCE, False, CHUNK, args=ARGS,
                                 trace_secondary_rays=True, to_cpu="maps")
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - t1
        host_maps = {"rays_per_s": rays_per_frame / dt_h, "bytes_to_host": int(sum(v.numel() * v.element_size() for v in hm.values())),
                     "note": "same frame with the per-ray maps (rgb, depth, opacity, mask, normals, x_surface) copied to pageable host "
                             "memory chunk by chunk; the reference copies every dict entry incl. per-sample tensors (eval.py:735-736)"}
        hm = None

    train = None
    if not a.no_train:
        from mirror_nerf_amd import training
        train = training.synthetic_train_bench(dev, rays, steps=10, warmup=3, batch=1024)
        train_total = training.synthetic_train_bench(dev, rays, steps=10, warmup=3, batch=1024, loss_name="total")
        train["with_total_loss"] = {k: train_total[k] for k in ("value", "ms_per_step", "loss", "loss_fn")}

    hash_grid = hash_grid_leg(dev, rays) if (rank == 0 and world == 1 and not a.no_train) else None

    if rank == 0:
        split = a.precision == "split"
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        res = {
            "metric": "rendered rays/sec (primary+reflected)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 carried as hi/lo f16 pairs (f16 MFMA, f32 accumulate; max |diff| to the fp32 chain 3e-6)"
                     if split else "f32",
            "data": "synthetic",
            "config": {"workload": "eval.batched_inference 800x800, 64 coarse (sigma-only) + 192 fine samples/ray, "
                                   "chunk 32768, 1 reflection bounce, all-mirror mask: 640000 primary + 640000 "
                                   "reflected rays per frame per GPU; random-init 8x256 MirrorNeRF pair, seed 0",
                       "rays_per_step_per_gpu": rays_per_frame, "parallelism": f"{world} x independent frames"},
            "samples_per_s": value * evals_per_ray,
            "field_evals_per_ray": evals_per_ray,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "peak_note": ("dense f16 MFMA peak; `achieved` counts ALGORITHMIC fp32 FLOPs (1 318 912 per sample), "
                                       "the kernel executes 3 f16 products per fp32 product") if split else "fp32 MFMA peak",
                         "executed_tflops": achieved * (SPLIT_EXECUTED_FLOP_FULL / MN.FLOP_FULL if split else 1339392 / MN.FLOP_FULL),
                         "executed_frac": achieved * (SPLIT_EXECUTED_FLOP_FULL / MN.FLOP_FULL if split else 1339392 / MN.FLOP_FULL) / peak,
                         "vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": PMC_HBM_BYTES_PER_FULL_SAMPLE * (sum(B for B, _ in full) / max(1, len(full))),
                         "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (profiles/), "
                                         "not re-measured in this run; algorithmic bytes are 36 B/sample",
                         "kernel": ("mnrf::h2::field_split_kernel<false,false>" if split else "mnrf::s2::field_kernel<false,false>")
                                   + " (full 4-head evaluation, fine pass)",
                         "other_precision": {"precision": other, "achieved": other_tf,
                                             "peak": PEAK_FP32_MFMA_TFLOPS if split else PEAK_F16_MFMA_TFLOPS,
                                             "frac": other_tf / (PEAK_FP32_MFMA_TFLOPS if split else PEAK_F16_MFMA_TFLOPS),
                                             "avg_launch_ms": other_ms, "rays_per_s": other_rays,
                                             "note": "one frame of the same workload with the other arithmetic"},
                         "avg_launch_ms": ms_full / max(1, len(full)), "launches": len(full),
                         "flop_per_sample": MN.FLOP_FULL,
                         "sigma_only_kernel_tflops": sig_tf,
                         "field_kernel_time_fraction": (ms_full + ms_sig) * 1e-3 / dt},
        }
        if host_maps is not None:
            res["with_host_maps"] = host_maps
        if train is not None:
            res["train_step"] = train
        if hash_grid is not None:
            res["hash_grid_variant"] = hash_grid
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sds, a.cpu_rays)
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
