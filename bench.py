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

Multi-GPU (`--gpus N`): one process per GPU over RCCL.  Launched by `python -m torch.distributed.run ...
bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) -- or plainly as `python bench.py
--gpus N`, in which case it re-executes itself under torch.distributed.run on 127.0.0.1.  Frames are
independent units: every rank renders its own frame (weak scaling), no collective on the data path,
barrier + max-over-ranks timing; value = rays of all ranks / slowest rank's time.  With more than one
rank the line also carries `strong_scaling` (ONE frame dealt to the ranks in interleaved 4096-ray tiles,
dist.render_sharded, + the optional gather of the maps to rank 0) and `train_step` (the 1024-ray
training step per rank with one flat RCCL all-reduce of the 5.3 MB gradient, train.py:577-584).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # same guide, dense f16/bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
# MFMA FLOPs the split kernel EXECUTES per full sample: 1308 hi/lo tile pairs x 3 products x 2 groups x 4 waves x
# 16384 FLOP per v_mfma_f32_16x16x32_f16 / 128 samples (= 3 x the padded fp32 count; algorithmic: MN.FLOP_FULL); the same
# per sample with 3 groups per wave and 192 samples per workgroup
SPLIT_EXECUTED_FLOP_FULL = 1308 * 3 * 2 * 4 * 16384 // 128
# roofline.traffic: HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes of THIS command
# (scripts/pmc_passes.sh -> profiles/traffic.json, keyed by kernel name; FETCH_SIZE doubled per the guide's gfx950
# correction + WRITE_SIZE).  PMC counters cannot be read from inside the run: a kernel without an entry reports null.
from mirror_nerf_amd.benchlegs import (ARGS, CHUNK, H, N_IMPORTANCE, N_SAMPLES, W, _pmc, _traffic,  # noqa: E402,F401
                                       clustered_balance, hash_grid_leg, roughness_leg, trained_leg)


def build_models(dev):
    import mirror_nerf_amd as M
    from mirror_nerf_amd import synthetic as SY
    models, sds = SY.build_models(dev, SY.ALL_MIRROR, seed=0)
    return models, sds, {"xyz": M.Embedding(10), "dir": M.Embedding(4)}



def _physical_cores():
    """Physical cores of the host (the CPU baseline runs on a FIXED thread count = physical cores): distinct
    (package, core) pairs of /proc/cpuinfo, else the logical count."""
    try:
        seen, pkg = set(), "0"
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pkg = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    seen.add((pkg, line.split(":")[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def _cpu_reps(run, sample, budget_s, reps=3, trial_n=256, cap=16384):
    """run(rays) -> (seconds, reflected rays).  One warm-up, a trial that sizes the sample for ~budget_s / reps seconds per
    repetition, then `reps` timed repetitions of the SAME sample -> (n, reflected, [seconds...])."""
    run(sample(64))                                   # warm-up (thread pools, allocator)
    dt, _ = run(sample(trial_n))
    n = int(min(cap, max(trial_n, budget_s / reps / dt * trial_n)))
    n -= n % 64
    r = sample(n)
    times, traced = [], 0
    for _ in range(reps):
        dt, traced = run(r)
        times.append(dt)
    return n, traced, times


def _rate_stats(n_rays, times):
    rates = sorted(n_rays / t for t in times)
    return {"median": rates[len(rates) // 2], "min": rates[0], "max": rates[-1], "repetitions": len(rates)}


def _pick_threads(run, sample):
    """The thread count torch's CPU ops run this path fastest at, from a short trial over 8 / 16 / 32 / 64 / all physical cores
    (the elementwise sin / cat / relu passes are memory-bound and the GEMMs small: more threads than that SLOW it down --
    128 threads measured 7x slower than 16 on the GPU box's host)."""
    import torch
    from threadpoolctl import threadpool_limits
    phys = _physical_cores()
    best = None
    for t in sorted({min(phys, c) for c in (8, 16, 32, 64, phys)}):
        torch.set_num_threads(t)
        with threadpool_limits(limits=t):
            run(sample(64))
            dt = min(run(sample(256))[0] for _ in range(2))
        if best is None or dt < best[1]:
            best = (t, dt)
    return best[0]


def cpu_baseline(sds, budget_s=20.0):
    """The reference's path on the host cores, on a bounded sample of the same workload (primary rays spread over the frame +
    their reflected rays): oracle/torch_port.py -- plain torch CPU ops in the reference's own op structure (cat + linear per
    layer, chunk 32768, cumprod, searchsorted, sort).  BASELINE.md section 4: one warm-up, THREE repetitions of one sample, the
    median reported with min / max (rounds 1-4 reported a single shot: 405 / 783 / 1312 / 814 rays/s over four rounds).
    `value` runs at the thread count a short trial finds fastest (a baseline should be the host's best); `all_physical_cores`
    is the same sample at torch.set_num_threads(physical cores), the literal wording of BASELINE.md.  `numpy_oracle`: the
    bit-careful checker of the parity tests, one repetition, for the record."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import mirror_nerf_oracle as O
    from oracle import torch_port as TP
    phys = _physical_cores()
    rays = O.synthetic_rays(H, W)

    def sample(n):
        return rays[np.linspace(0, rays.shape[0] - 1, n).astype(np.int64)]

    mt = {k: {n: torch.from_numpy(v) for n, v in sd.items()} for k, sd in zip(("coarse", "fine"), sds)}

    def run_torch(r):
        t0 = time.perf_counter()
        out = TP.render_eval(mt, torch.from_numpy(np.ascontiguousarray(r)), N_SAMPLES, N_IMPORTANCE, CHUNK, 1)
        return time.perf_counter() - t0, int((out["mirror_mask_fine"] != 0).any()) * r.shape[0]

    def run_numpy(r):
        t0 = time.perf_counter()
        out = O.render_eval({"coarse": sds[0], "fine": sds[1]}, {"xyz": 10, "dir": 4}, r, N_SAMPLES, N_IMPORTANCE, False,
                            CHUNK, ARGS)
        return time.perf_counter() - t0, int((out["mirror_mask_fine"] != 0).any()) * r.shape[0]

    threads = _pick_threads(run_torch, sample)
    torch.set_num_threads(threads)
    with threadpool_limits(limits=threads):
        n_t, tr_t, times_t = _cpu_reps(run_torch, sample, 0.6 * budget_s, cap=4096)
        O.set_sgemm(lambda x, w: torch.mm(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(w).t()).numpy())
        try:
            n_n, tr_n, times_n = _cpu_reps(run_numpy, sample, 0.2 * budget_s, reps=1)
        finally:
            O.set_sgemm(None)
    st = _rate_stats(n_t + tr_t, times_t)
    res = {"value": st["median"], "unit": "rays/s", "cores": threads, "kind": "port", "min": st["min"], "max": st["max"],
           "repetitions": st["repetitions"], "physical_cores": phys, "logical_cpus": os.cpu_count(),
           "sample": f"{n_t} primary + {tr_t} reflected rays of the same frame, oracle/torch_port.py (plain torch CPU ops in the "
                     f"reference's op structure, chunk {CHUNK}), {threads} threads (the fastest of a trial over 8..{phys}), one warm-up "
                     f"+ {len(times_t)} repetitions of {sum(times_t) / len(times_t):.1f} s: median (min / max beside it)",
           "numpy_oracle": {"value": (n_n + tr_n) / times_n[0], "unit": "rays/s", "cores": threads,
                            "sample": f"{n_n} primary + {tr_n} reflected rays, oracle/mirror_nerf_oracle.py with torch's CPU "
                                      f"sgemm as its GEMM backend ({threads} threads), one repetition of {times_n[0]:.1f} s"}}
    if phys != threads:
        torch.set_num_threads(phys)
        with threadpool_limits(limits=phys):
            n_p, tr_p, times_p = _cpu_reps(run_torch, sample, 0.2 * budget_s, reps=1, cap=4096)
        res["all_physical_cores"] = {"value": (n_p + tr_p) / times_p[0], "unit": "rays/s", "cores": phys,
                                     "sample": f"{n_p} primary + {tr_p} reflected rays, torch.set_num_threads({phys}), one repetition"}
        torch.set_num_threads(threads)
    return res


def config1_cpu(sd_coarse, budget_s=8.0, threads=None):
    """BASELINE config 1 as worded -- 400x400, coarse-only 64 samples, one bounce, "PyTorch CPU path (plumbing, no GPU)" -- on the
    host cores: the torch port under TRAIN semantics with a ground-truth mirror mask (the centred 25 % rectangle of SURVEY 8d),
    a bounded sample of the 160 000 + 40 000 rays, three repetitions, median."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import mirror_nerf_oracle as O
    from oracle import torch_port as TP
    threads = threads or _physical_cores()      # (bench.main passes the count cpu_baseline's trial picked)
    Hc = Wc = 400
    rays = O.synthetic_rays(Hc, Wc)
    gt = np.zeros((Hc, Wc), np.float32)
    gt[Hc // 4: Hc - Hc // 4, Wc // 4: Wc - Wc // 4] = 1.0
    gt = gt.reshape(-1)
    mt = {"coarse": {n: torch.from_numpy(v) for n, v in sd_coarse.items()}}

    def sample(n):
        return np.linspace(0, rays.shape[0] - 1, n).astype(np.int64)

    def run(idx):
        r, m = torch.from_numpy(np.ascontiguousarray(rays[idx])), torch.from_numpy(np.ascontiguousarray(gt[idx]))
        t0 = time.perf_counter()
        TP.render_train_coarse(mt, r, m, N_SAMPLES, CHUNK)
        return time.perf_counter() - t0, int(m.sum())
    torch.set_num_threads(threads)
    with threadpool_limits(limits=threads):
        run(sample(64))
        dt, _ = run(sample(512))
        n = int(min(65536, max(512, budget_s / 3 / dt * 512)))
        idx = sample(n - n % 64)
        times, refl = [], 0
        for _ in range(3):
            dt, refl = run(idx)
            times.append(dt)
    st = _rate_stats(len(idx) + refl, times)
    return {"value": st["median"], "min": st["min"], "max": st["max"], "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{len(idx)} primary + {refl} reflected rays spread over the 400x400 frame, coarse-only 64 samples, "
                      f"oracle/torch_port.render_train_coarse, {threads} threads, 3 repetitions of {sum(times) / 3:.1f} s"}


def _respawn(a):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def _build_commit():
    try:
        with open(os.path.join(ROOT, "mirror_nerf_amd", "BUILD_COMMIT")) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--no-train", action="store_true", help="skip the short training-step measurement")
    ap.add_argument("--precision", choices=("split", "fp32"), default="split",
                    help="arithmetic of the field kernel's Linears: fp32 operands as hi/lo f16 pairs on the f16 matrix "
                         "pipe (default, ~1e-6 of fp32) or the bit-exact fp32 MFMA chain")
    a = ap.parse_args()
    # MNRF_BENCH_LEGS=headline,strong,other,host_maps,fused,train,train_total,config3,hash_grid,rough,trained,config1,cpu : run only
    # these secondary legs (default: all that apply) -- an 8-rank run can be bounded to "headline,strong,train"
    want = os.environ.get("MNRF_BENCH_LEGS")
    want = None if not want else {w.strip() for w in want.split(",") if w.strip()}
    leg = (lambda name: True) if want is None else (lambda name: name in want)

    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("MNRF_BENCH_SPAWN") == "1"):
        _respawn(a)          # MNRF_BENCH_SPAWN=1: take the launcher path at N = 1 too (tests on a 1-GPU box)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus}, or plain "
                 f"python bench.py --gpus {a.gpus})")

    # a rank that stops making progress says where: every thread's stack goes to stderr after MNRF_BENCH_WATCHDOG seconds
    # (default 900; the run is not interrupted)
    import faulthandler
    wd = float(os.environ.get("MNRF_BENCH_WATCHDOG", "900"))
    if wd > 0:
        faulthandler.dump_traceback_later(wd, repeat=False, file=sys.stderr)

    import torch
    import torch.distributed as dist
    import mirror_nerf_amd as M
    from mirror_nerf_amd import dist as D
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY

    rank, world, dev = D.init_from_env()
    if dist.is_initialized() and dist.get_world_size() != a.gpus:
        sys.exit(f"bench.py: RCCL group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
    multi = dist.is_initialized()

    MN.set_precision(a.precision)
    models, sds, emb = build_models(dev)
    # every rank renders its own view: same camera model, pose rotated about z by the rank index
    rays = SY.device_rays(H, W, dev, SY.look_at_pose(eye=(4.0 * np.sin(0.3 * rank), -4.0 * np.cos(0.3 * rank), 1.5)))

    def render(r, to_cpu=False):
        return M.batched_inference(models, emb, r, N_SAMPLES, N_IMPORTANCE, False, CHUNK, args=ARGS,
                                   trace_secondary_rays=True, to_cpu=to_cpu)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        out = render(rays)
    sync()
    MN.LAUNCH_LOG = []
    # clock / power / temperature / throttle state of this rank's GPU, sampled every 200 ms THROUGH the timed frames
    # (mirror_nerf_amd/telemetry.py): the dominant kernel runs at the package power limit, so the clock the box grants sets
    # its speed -- the line must be able to tell a slow box from slow code
    from mirror_nerf_amd.telemetry import SmiSampler
    smi_main = SmiSampler(dev.index or 0, 0.2)
    with smi_main:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = render(rays)
        sync()
        dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    log, MN.LAUNCH_LOG = MN.LAUNCH_LOG, None

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

    # everything the line is assembled from, so that emit() can run at any point from here on
    strong = other_tf = other_rays = other_ms = smi_other = host_maps = fused = train = hash_grid = rough = trained = config1 = None
    legs_done = []

    def emit(incomplete=None, early=False):
        """Rank 0 prints the line.  N = 1: ONE line, at the end.  N > 1: the headline line goes out RIGHT AFTER the weak-scaling
        measurement (`"line": "headline"`, flushed), the complete one replaces it at the end (`"line": "complete"`): a harness
        takes the last parseable line, and a stall in a secondary leg cannot cost the run its number.  `incomplete`: legs cut
        short by the deadline below."""
        if rank == 0:
            split = a.precision == "split"
            peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
            h = "h2" if os.environ.get("MNRF_SPLIT48", "1") == "0" else "h3"     # 48 samples per wave is the default tuning
            kernel = f"mnrf::{h}::field_split_kernel<false,false,false,false>" if split else "mnrf::s2::field_kernel<false,false>"
            traffic, traffic_src, traffic_commit = _traffic(kernel)
            tele = smi_main.summary()
            sclk = smi_main.median_sclk()
            # the peak the kernel could reach at the clock this box granted it (peak is quoted at 2.4 GHz)
            granted_peak = peak * sclk / 2400.0 if sclk else None
            res = {
                "metric": "rendered rays/sec (primary+reflected)", "value": value, "unit": "rays/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 carried as hi/lo f16 pairs (f16 MFMA, f32 accumulate; max |diff| to the fp32 chain 3e-6)"
                         if split else "f32",
                "data": "synthetic", "code_commit": _build_commit(),
                "config": {"workload": "eval.batched_inference 800x800, 64 coarse (sigma-only) + 192 fine samples/ray, "
                                       "chunk 32768, 1 reflection bounce, all-mirror mask: 640000 primary + 640000 "
                                       "reflected rays per frame per GPU; random-init 8x256 MirrorNeRF pair, seed 0",
                           "rays_per_step_per_gpu": rays_per_frame, "parallelism": f"{world} x independent frames",
                           "train_step_workload": "1024 rays per GPU, perturb = noise_std = 1, 25 % GT mirror rays reflected once; train_step: "
                                                  "run.sh:266's schedule (64 coarse + 128 fine samples, --N_importance 64); "
                                                  "train_step.config3_64_plus_192: BASELINE config 3 as worded (64 + 192, --N_importance 128)",
                           "collective_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (not RCCL: timings void)")) if multi else None,
                           "rccl_world_size": dist.get_world_size() if multi else None},
                "samples_per_s": value * evals_per_ray,
                "field_evals_per_ray": evals_per_ray,
                "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                             "frac": achieved / peak,
                             "peak_note": ("dense f16 MFMA peak; `achieved` counts ALGORITHMIC fp32 FLOPs (1 318 912 per sample), "
                                           "the kernel executes 3 f16 products per fp32 product") if split else "fp32 MFMA peak",
                             "executed_tflops": achieved * (SPLIT_EXECUTED_FLOP_FULL / MN.FLOP_FULL if split else 1339392 / MN.FLOP_FULL),
                             "executed_frac": achieved * (SPLIT_EXECUTED_FLOP_FULL / MN.FLOP_FULL if split else 1339392 / MN.FLOP_FULL) / peak,
                             "vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                             "frac_at_granted_clock": achieved / granted_peak if granted_peak else None,
                             "granted_clock_note": "frac with `peak` scaled to the median shader clock sampled through the timed frames "
                                                   "(telemetry.sclk_mhz.median / 2400 MHz): separates what the box granted from what the code does",
                             "traffic": traffic,
                             "traffic_source": (f"static profile: profiles/traffic.json, measured at commit {traffic_commit or 'unrecorded'} "
                                                f"by {traffic_src}") if traffic is not None else None,
                             "traffic_note": ("HBM bytes per launch = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE from separate rocprofv3 "
                                              "--pmc passes over one chunk of this workload (PMC counters cannot be read inside this run; "
                                              "scripts/profile_round.sh regenerates the file); algorithmic bytes are 36 B/sample")
                                             if traffic is not None else "no PMC pass on record for this kernel (profiles/traffic.json)",
                             "telemetry": tele,
                             "kernel": kernel + " (full 4-head evaluation, fine pass)",
                             "avg_launch_ms": ms_full / max(1, len(full)), "launches": len(full),
                             "flop_per_sample": MN.FLOP_FULL,
                             "sclk_mhz_under_load": (tele["sclk_mhz"] or {}).get("median"),
                             "board_power_w_under_load": (tele["power_w"] or {}).get("median"),
                             "board_power_cap_w": tele["power_cap_w"],
                             "power_note": "medians of the 200 ms samples taken through the timed frames (`telemetry` has min/median/max, "
                                           "throttle bits and the power-limit residency); the split-f16 kernels run at the package power "
                                           "limit and get ~2.1-2.2 GHz instead of the 2.4 GHz `peak` assumes (profiles/DIARY.md 9.1)",
                             "sigma_only_kernel_tflops": sig_tf,
                             "field_kernel_time_fraction": (ms_full + ms_sig) * 1e-3 / dt},
            }
            if other_tf is not None:
                res["roofline"]["other_precision"] = {
                    "precision": other, "achieved": other_tf, "peak": PEAK_FP32_MFMA_TFLOPS if split else PEAK_F16_MFMA_TFLOPS,
                    "frac": other_tf / (PEAK_FP32_MFMA_TFLOPS if split else PEAK_F16_MFMA_TFLOPS),
                    "avg_launch_ms": other_ms, "rays_per_s": other_rays, "note": "one frame of the same workload with the other arithmetic",
                    "telemetry": smi_other.summary() if smi_other is not None else None,
                    "frac_at_granted_clock": (other_tf / ((PEAK_FP32_MFMA_TFLOPS if split else PEAK_F16_MFMA_TFLOPS) * smi_other.median_sclk() / 2400.0)
                                              if (smi_other is not None and smi_other.median_sclk()) else None)}
                res["fp32_rays_per_s" if split else "split_rays_per_s"] = other_rays
                res["fp32_frac" if split else "split_frac"] = res["roofline"]["other_precision"]["frac"]
            if strong is not None:
                res["strong_scaling"] = strong
            if host_maps is not None:
                res["with_host_maps"] = host_maps
            if fused is not None:
                res["maps_only_fused"] = fused
            if train is not None:
                res["train_step"] = train
                # top level, so that a harness that keeps only the keys of nested objects keeps the numbers:
                # ms per step of [colour + mask | TotalLoss | config-3 schedule 64 + 192 | run.sh recipe after / inside its geometry stage]
                def g_(*ks):
                    d = train
                    for k in ks:
                        d = d.get(k) if isinstance(d, dict) else None
                    return d.get("ms_per_step") if isinstance(d, dict) else None
                res["train_ms"] = [train.get("ms_per_step"), g_("with_total_loss"), g_("config3_64_plus_192"),
                                   g_("run_sh_recipe", "after_geometry_stage"), g_("run_sh_recipe", "in_geometry_stage")]
                res["train_route"] = train.get("route")
                res["train_frac"] = (train.get("roofline") or {}).get("frac")
            if hash_grid is not None:
                res["hash_grid_variant"] = hash_grid
            if rough is not None:
                res["roughness_variant"] = rough
            if trained is not None:
                res["trained_weights_variant"] = trained
            if config1 is not None:
                res["config1"] = config1
            if world == 1 and not a.no_cpu_baseline and leg("cpu") and not early:
                res["cpu_baseline"] = cpu_baseline(sds, a.cpu_seconds)
                if config1 is not None:
                    config1["cpu_baseline"] = config1_cpu(sds[0], threads=res["cpu_baseline"]["cores"])
            if incomplete:
                res["incomplete_legs"] = incomplete
            if multi:
                res["line"] = "headline" if early else "complete"
            if want is not None:
                res["legs_requested"] = sorted(want)
            print(json.dumps(res), flush=True)


    # N > 1: the legs after the headline measurement are bounded.  A rank that stalls in a secondary leg (a collective that never
    # completes) must not cost the run its line: after MNRF_BENCH_LEG_DEADLINE seconds (default 900) rank 0 prints what was
    # measured, names the legs that did not finish, and every rank leaves.  `value` was measured above, before any of this.
    deadline = None
    if multi:
        import threading

        def _expired():
            try:
                faulthandler.dump_traceback(file=sys.stderr)
                emit(incomplete={"finished": list(legs_done), "note": "secondary legs cut by MNRF_BENCH_LEG_DEADLINE; "
                                 "value / roofline were measured before them"})
            finally:
                sys.stdout.flush()
                # non-zero: a harness that keys on the return code must not record a cut run (a collective that never
                # completed) as a pass; the headline line is on stdout all the same (MNRF_BENCH_DEADLINE_RC overrides)
                os._exit(int(os.environ.get("MNRF_BENCH_DEADLINE_RC", "3")))
        deadline = threading.Timer(float(os.environ.get("MNRF_BENCH_LEG_DEADLINE", "900")), _expired)
        deadline.daemon = True
        deadline.start()
        emit(early=True)

    # strong scaling: ONE frame (rank 0's view) dealt to the ranks in interleaved 4096-ray tiles; no collective while
    # rendering; then the optional assembly of the 20 B/ray maps on rank 0
    strong = None
    if multi and leg("strong"):
        common = SY.device_rays(H, W, dev)
        keys = ("rgb_fine", "depth_fine", "mirror_mask_fine")
        idx, res = D.render_sharded(render, common)                       # warm-up
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            idx, res = D.render_sharded(render, common)
        sync()
        dt_s = D.max_over_ranks(time.perf_counter() - t1, dev)
        t1 = time.perf_counter()
        frame = D.gather_frame(idx, res, H * W, keys=keys)
        sync()
        dt_g = D.max_over_ranks(time.perf_counter() - t1, dev)
        strong = {"rays_per_s": 2 * H * W * a.steps / dt_s, "ms_per_frame": dt_s / a.steps * 1e3, "tile": D.TILE,
                  "rays_of_rank0": int(idx.numel()), "gather_ms": dt_g * 1e3,
                  "gathered_bytes": int(sum(v.numel() * 4 for v in frame.values())) if rank == 0 else None,
                  "note": "one 800x800 frame + its reflected rays over all ranks (interleaved 4096-ray tiles, no data-path "
                          "collective); gather = all_gather of rgb/depth/mask maps, outside ms_per_frame"}
        res = frame = None
        strong["clustered_mask"] = clustered_balance(dev, models, common, rank, world, sync)
        legs_done.append("strong_scaling")

    # the other arithmetic on ONE whole frame of the same workload, for the record: rays/s and the dominant kernel's rate
    other = "fp32" if a.precision == "split" else "split"
    other_tf = other_rays = other_ms = smi_other = None
    host_maps = None
    if world == 1 and leg("other"):
        MN.set_precision(other)
        render(rays)                 # warm-up (first launches of the other kernels)
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        smi_other = SmiSampler(dev.index or 0, 0.2)
        with smi_other:
            t1 = time.perf_counter()
            render(rays)
            torch.cuda.synchronize()
            dt_o = time.perf_counter() - t1
        t_o = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in MN.LAUNCH_LOG if not (flags & 1)]
        other_tf = sum(B for B, _ in t_o) * MN.FLOP_FULL / (sum(t for _, t in t_o) * 1e-3) / 1e12
        other_rays = (H * W * 2) / dt_o
        other_ms = sum(t for _, t in t_o) / max(1, len(t_o))
        MN.LAUNCH_LOG = None
        MN.set_precision(a.precision)
    if world == 1 and leg("host_maps"):
        # PCIe-inclusive rate (never `value`): frames whose per-ray maps are copied to the host (one untimed first: it pins the
        # staging buffers, as the first frame of an eval run does)
        render(rays, to_cpu="maps")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hm = render(rays, to_cpu="maps")
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - t1
        host_maps = {"rays_per_s": rays_per_frame / dt_h, "bytes_to_host": int(sum(v.numel() * v.element_size() for v in hm.values())),
                     "note": "same frame with the per-ray maps (rgb, depth, opacity, mask, normals, x_surface) handed to the host: chunk "
                             "by chunk into pinned staging buffers on a side stream while the next chunk renders, one host-side copy "
                             "at the end; final pass ray-fused (maps_only_fused is the same frame left on the device); the reference "
                             "copies every dict entry incl. per-sample tensors (eval.py:735-736)"}
        hm = None

    # the same frame with per-ray maps only, results left on the device: the final pass ray-fused (field evaluation +
    # compositing in one kernel, head outputs in LDS, no per-sample tensor in HBM: SURVEY 3 "result-dict contract")
    fused = None
    if world == 1 and leg("fused"):
        render_maps = lambda r: M.batched_inference(models, emb, r, N_SAMPLES, N_IMPORTANCE, False, CHUNK, args=ARGS,  # noqa: E731
                                                    trace_secondary_rays=True, to_cpu=False, maps_only=True)
        render_maps(rays)
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        t1 = time.perf_counter()
        for _ in range(2):
            render_maps(rays)
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t1) / 2
        t_f = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in MN.LAUNCH_LOG if flags & 0x2000]
        MN.LAUNCH_LOG = None
        fk = "mnrf::h3::field_split_kernel<false,false,false,true>"
        ftr, fsrc, fcommit = _traffic(fk)
        fused = {"rays_per_s": rays_per_frame / dt_f, "ms_per_frame": dt_f * 1e3,
                 "kernel": fk + " (full evaluation + compositing of one 192-sample ray per workgroup)",
                 "avg_launch_ms": sum(t for _, t in t_f) / max(1, len(t_f)), "launches": len(t_f),
                 "achieved_tflops": sum(B for B, _ in t_f) * MN.FLOP_FULL / (sum(t for _, t in t_f) * 1e-3) / 1e12 if t_f else None,
                 "traffic": ftr, "traffic_source": (f"static profile: profiles/traffic.json, commit {fcommit or 'unrecorded'}, {fsrc}") if ftr else None,
                 "algorithmic_bytes_per_launch": CHUNK * (N_SAMPLES + N_IMPORTANCE) * 4 + CHUNK * (32 + 48),
                 "note": "batched_inference(..., to_cpu=False, maps_only=True): identical maps bit for bit (tests), the per-sample keys "
                         "of the final pass are not produced; to_cpu=\"maps\" (with_host_maps) takes the same kernels"}

    train = None
    # More than one rank: the contract's legs run on the STATIC route (host-issued launches, host-issued collectives: the path every
    # multi-rank test has exercised); the captured step WITH its collectives (round 6) is measured after the complete line is out, as an
    # optional last leg -- it has only ever run on a 1-rank RCCL group, and a collective that cannot be captured on some stack must
    # not cost the run its line.  MNRF_TRAIN_ROUTE overrides.
    rk = {"_route": "static"} if (multi and not os.environ.get("MNRF_TRAIN_ROUTE")) else {}
    if not a.no_train and leg("train"):
        from mirror_nerf_amd import training
        train = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, **rk)
        legs_done.append("train_step")
        if world == 1 and leg("train_routes"):
            # the same step on the reference's shape (the host reads the reflected-ray count in the middle of the step,
            # train.py:175) and as host-issued static launches: what the round-5 routes are measured against
            train["routes_ms_per_step"] = {train["route"]: train["ms_per_step"]}
            for r_ in ("static", "host"):
                t_ = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, _route=r_)
                train["routes_ms_per_step"][r_] = t_["ms_per_step"]
        if leg("train_total"):
            train_total = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, loss_name="total", **rk)
            train["with_total_loss"] = {k: train_total[k] for k in ("value", "ms_per_step", "loss", "loss_fn", "roofline", "route")
                                        if k in train_total}
        if leg("config3"):
            # BASELINE config 3 as worded ("same config" as config 2: 64 coarse + 128 importance samples); the default above is
            # run.sh:266's training schedule (--N_importance 64)
            train_c3 = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, N_importance=128, **rk)
            train["config3_64_plus_192"] = {k: train_c3[k] for k in ("value", "ms_per_step", "samples_per_ray", "N_importance", "roofline",
                                                                      "reflected_rays_per_step", "allreduce", "route")}
        if world == 1 and leg("half_planes"):
            # opt-in, NOT exact (MNRF_DW_PLANES_HALF; include/mnrf.h MNRF_PLANES_Y_HALF): dY reaches the weight-gradient GEMM as one
            # f16 per element.  Reported beside the default, never instead of it.
            from mirror_nerf_amd import autograd as AG
            AG.DW_PLANES_HALF = True
            try:
                t_ = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024)
            finally:
                AG.DW_PLANES_HALF = False
            train["opt_in_half_dy_planes"] = {
                "ms_per_step": t_["ms_per_step"], "value": t_["value"], "route": t_["route"], "roofline": t_["roofline"],
                "note": "MNRF_DW_PLANES_HALF=1: the activation gradients travel to the weight-gradient GEMM as ONE f16 per element under "
                        "the planes' per-sample scale (the producer's lo tiles are dropped by a zero-record buffer descriptor, the GEMM "
                        "fetches and multiplies hi tiles only: 3/4 of its bytes).  Not exact: within 1e-3 of each weight tensor's largest "
                        "entry of the default route on G9 / G16 / G11 (tests/test_hip_backward.py::test_half_dy_planes_hold_the_gradient_bar), "
                        "1.2e-4 .. 7.4e-4 against float64 in the emulation on the reference (profiles/r06_half_planes_emulation.json)"}
        if leg("run_sh"):
            # run.sh:259-280 as the reference trains: --use_plane_consistent_loss --train_geometry_stage.  All five terms of TotalLoss
            # (the plane term drawing on the device), after the geometry stage (reflections traced) and inside it (none), on the
            # default route (the captured graph)
            rs = {}
            for key, name in (("after_geometry_stage", "run_sh"), ("in_geometry_stage", "run_sh_stage")):
                t_ = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, loss_name=name, **rk)
                rs[key] = {k: t_[k] for k in ("value", "ms_per_step", "loss", "loss_fn", "roofline", "route", "reflected_rays_per_step",
                                             "collectives_in_graph", "allreduce") if k in t_}
            train["run_sh_recipe"] = rs
        legs_done.append("train_step.with_total_loss + config3_64_plus_192 + run_sh_recipe")

    hash_grid = hash_grid_leg(dev, rays) if (not a.no_train and leg("hash_grid")) else None
    legs_done.append("hash_grid_variant")
    rough = roughness_leg(dev, models, emb) if (world == 1 and not a.no_train and leg("rough")) else None
    trained = trained_leg(dev) if (world == 1 and not a.no_train and leg("trained")) else None
    if world == 1 and not a.no_train and leg("config1"):
        from mirror_nerf_amd.benchlegs import config1_leg
        config1 = config1_leg(dev)

    if deadline is not None:
        deadline.cancel()
    emit()
    if multi and train is not None and not os.environ.get("MNRF_TRAIN_ROUTE") and os.environ.get("MNRF_BENCH_GRAPH_ACROSS_RANKS", "1") != "0" \
            and dist.get_backend() == "nccl":
        # optional last leg (see above): the step captured with its RCCL collectives inside.  The complete line is out; if this leg
        # does not finish within MNRF_BENCH_GRAPH_DEADLINE seconds (default 300) the process leaves with code 0 and says so on stderr.
        import threading

        def _give_up():
            sys.stderr.write("bench.py: the optional leg train_step.graph_across_ranks did not finish (the complete line above stands)\n")
            sys.stderr.flush()
            os._exit(0)
        t_opt = threading.Timer(float(os.environ.get("MNRF_BENCH_GRAPH_DEADLINE", "300")), _give_up)
        t_opt.daemon = True
        t_opt.start()
        from mirror_nerf_amd import training
        g_ = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, _route="graph")
        t_opt.cancel()
        train["graph_across_ranks"] = {k: g_[k] for k in ("value", "ms_per_step", "route", "collectives_in_graph", "allreduce", "roofline") if k in g_}
        train["graph_across_ranks"]["note"] = ("the colour + mask step as ONE hipGraph per rank with its collectives inside (bucket all-reduces from "
                                               "the backward hooks, guard words OR-ed over the ranks); `route` says \"static (...)\" if the capture "
                                               "failed on this stack and the leg fell back")
        emit()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
