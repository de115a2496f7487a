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

H = W = 800
N_SAMPLES, N_IMPORTANCE, CHUNK = 64, 128, 32768
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # same guide, dense f16/bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
# MFMA FLOPs the split kernel EXECUTES per full sample: 1308 hi/lo tile pairs x 3 products x 2 groups x 4 waves x
# 16384 FLOP per v_mfma_f32_16x16x32_f16 / 128 samples (= 3 x the padded fp32 count; algorithmic: MN.FLOP_FULL); the same
# per sample with 3 groups per wave and 192 samples per workgroup
SPLIT_EXECUTED_FLOP_FULL = 1308 * 3 * 2 * 4 * 16384 // 128
# roofline.traffic: HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes of THIS command
# (scripts/pmc_passes.sh -> profiles/traffic.json, keyed by kernel name; FETCH_SIZE doubled per the guide's gfx950
# correction + WRITE_SIZE).  PMC counters cannot be read from inside the run: a kernel without an entry reports null.
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


def build_models(dev):
    import mirror_nerf_amd as M
    from mirror_nerf_amd import synthetic as SY
    models, sds = SY.build_models(dev, SY.ALL_MIRROR, seed=0)
    return models, sds, {"xyz": M.Embedding(10), "dir": M.Embedding(4)}


ARGS = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)


def cpu_baseline(sds, budget_s=20.0):
    """The reference's path on the host cores, on a bounded sample of the same workload (primary rays spread over
    the frame + their reflected rays), two ways:
      * `value`: oracle/torch_port.py -- plain torch CPU ops in the reference's own op structure (cat + linear per
        layer, chunk 32768, cumprod, searchsorted, sort), torch.set_num_threads(threads): what the reference's CPU
        path costs here;
      * `numpy_oracle`: oracle/mirror_nerf_oracle.py, the bit-careful checker the parity tests use, for the record.
    The sample is sized from a short trial so that each leg takes about `budget_s`/2 seconds."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import mirror_nerf_oracle as O
    from oracle import torch_port as TP
    ncpu = os.cpu_count() or 1
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

    # torch's CPU ops do not scale to every core of a big host (the elementwise sin / cat / relu passes of this path
    # are memory-bound and the GEMMs are small): try a few thread counts on a short trial and keep the fastest
    best = None
    for t in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(t)
        with threadpool_limits(limits=t):
            run_torch(sample(64))
            dt, _ = run_torch(sample(256))
        if best is None or dt < best[1]:
            best = (t, dt)
    threads = best[0]
    torch.set_num_threads(threads)

    def sized(run, trial_n=256):
        run(sample(64))                                   # warm-up (thread pools, allocator)
        dt, _ = run(sample(trial_n))
        n = int(min(16384, max(trial_n, 0.5 * budget_s / dt * trial_n)))
        n -= n % 64
        dt, traced = run(sample(n))
        return n, traced, dt

    with threadpool_limits(limits=threads):
        n_t, tr_t, dt_t = sized(run_torch)
        O.set_sgemm(lambda x, w: torch.mm(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(w).t()).numpy())
        try:
            n_n, tr_n, dt_n = sized(run_numpy)
        finally:
            O.set_sgemm(None)
    return {"value": (n_t + tr_t) / dt_t, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{n_t} primary + {tr_t} reflected rays of the same frame, oracle/torch_port.py (plain torch CPU ops in "
                      f"the reference's op structure, chunk {CHUNK}, {threads} threads = the fastest of a trial over 8..{ncpu}), {dt_t:.1f} s",
            "numpy_oracle": {"value": (n_n + tr_n) / dt_n, "unit": "rays/s", "cores": threads,
                             "sample": f"{n_n} primary + {tr_n} reflected rays, oracle/mirror_nerf_oracle.py with torch's CPU "
                                       f"sgemm as its GEMM backend ({threads} threads), {dt_n:.1f} s"}}


def hash_grid_leg(dev, rays):
    """BASELINE config 5 for the record (not `value`): the hash-grid field (MirrorNeRFTcnn, bound 6, 2^19 x 16 x 2 table)
    on the same 800x800 rays -- one frame of primary rays through render_rays (64 sigma-only + 192 full samples), and the
    1024-ray training step (forward + hand-written backward + gradient all-reduce + Adam).  With N ranks (config 5 is worded
    "... 8xMI355X"): every rank renders the whole frame (weak scaling, no data-path collective) and the training step reduces
    the two table gradients in place and the MLP gradients as one blob per model (dist._module_messages)."""
    import torch
    import mirror_nerf_amd as M
    from mirror_nerf_amd import dist as D
    rank, world = D.world()
    collective = world > 1 or D.forced()
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
    from mirror_nerf_amd import mirror_nerf as MN
    MN.LAUNCH_LOG = []
    if collective:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    frame()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    dt_f = D.max_over_ranks(dt_own, dev)
    log, MN.LAUNCH_LOG = MN.LAUNCH_LOG, None
    full = [(B, e0.elapsed_time(e1)) for (flags, B, e0, e1) in log if (flags & 0x1000) and not (flags & 1)]
    ms_full = sum(t for _, t in full)
    # the dominant kernel of this variant is gather-bound: 16 levels x 8 corners x 8 B (float2) = 1 KiB of table reads per
    # sample, from a 53 MB table (Infinity-Cache resident: the HBM peak is the contract's yardstick, not the binding limit)
    gbs = sum(B for B, _ in full) * (1024 + 256) / (ms_full * 1e-3) / 1e9 if ms_full > 0 else 0.0
    # the binding limit: random gathers out of the Infinity Cache, measured on this very table (mnrf_bench_gather)
    from mirror_nerf_amd import _lib
    table = models["fine"].encoder.embeddings.detach()
    sink = torch.zeros(4, device=dev)
    ceil = {}
    for key, nbytes, span in (("random_8B", 8, table.numel() * 4), ("random_4B", 4, table.numel() * 4), ("l2_resident_8B", 8, 2 << 20)):
        n_thr, iters = 256 * 4096, 256
        _lib.check(_lib.lib().mnrf_bench_gather(_lib.ptr(table), span, nbytes, n_thr, iters, _lib.ptr(sink), _lib.stream()), "gather")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _lib.check(_lib.lib().mnrf_bench_gather(_lib.ptr(table), span, nbytes, n_thr, iters, _lib.ptr(sink), _lib.stream()), "gather")
        e1.record()
        torch.cuda.synchronize()
        ceil[key] = 3 * n_thr * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9       # G gathers / s
    gathers_per_s = sum(B for B, _ in full) * 128 / (ms_full * 1e-3) / 1e9 if ms_full > 0 else 0.0
    # the level-major encoding launch on its own (mnrf_tcnn_encode) and the whole field evaluation on the same fine-pass samples
    with torch.no_grad():
        rc = M.render_rays(models, emb, rays[:CHUNK], N_SAMPLES, False, 0, 0, N_IMPORTANCE, CHUNK, test_time=True, compute_normal=False)
    zf = rc["z_vals_fine"].contiguous()
    mfine = models["fine"]
    from mirror_nerf_amd.mirror_nerf_tcnn import _offsets17
    offs = _offsets17(mfine.cfg)
    planes = torch.empty(32 * zf.numel(), device=dev)
    rchunk = rays[:CHUNK].contiguous()
    pr = lambda: _lib.check(_lib.lib().mnrf_tcnn_encode(  # noqa: E731
        _lib.ptr(table), offs, mfine.cfg["S"], mfine.cfg["H"], float(mfine.bound), zf.numel(), None, 0,
        _lib.ptr(rchunk), _lib.ptr(zf), zf.shape[1], _lib.ptr(planes), _lib.stream()), "encode")
    pr()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pr()
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / 5
    planes = None
    MN.LAUNCH_LOG = []
    with torch.no_grad():
        for _ in range(3):
            mfine.field(zf.numel(), rays=rchunk, z_vals=zf, spr=zf.shape[1])
    torch.cuda.synchronize()
    same_ms = sum(a_.elapsed_time(b_) for (_f, _B, a_, b_) in MN.LAUNCH_LOG) / 3
    MN.LAUNCH_LOG = None
    one_launch_ms = None
    for m in models.values():
        m.enc_planes_min = 1 << 62
    MN.LAUNCH_LOG = []
    with torch.no_grad():
        for _ in range(3):
            mfine.field(zf.numel(), rays=rchunk, z_vals=zf, spr=zf.shape[1])
    torch.cuda.synchronize()
    one_launch_ms = sum(a_.elapsed_time(b_) for (_f, _B, a_, b_) in MN.LAUNCH_LOG) / 3
    MN.LAUNCH_LOG = None
    for m in models.values():
        del m.enc_planes_min
    pmc = _pmc("mnrf::mf::tcnn_encode_kernel")
    l2_bytes = pmc.get("TCP_TCC_READ_REQ_sum", 0.0) * 128.0 if pmc else None      # 128-byte lines requested from the L2 per launch
    # single-pass f16 MLPs ("fp16 MLP on CDNA4 MFMA", BASELINE config 5; module.mlp_f16): the frame again
    for m in models.values():
        m.mlp_f16 = True
    frame()
    torch.cuda.synchronize()
    MN.LAUNCH_LOG = []
    t0 = time.perf_counter()
    frame()
    torch.cuda.synchronize()
    dt_f16 = D.max_over_ranks(time.perf_counter() - t0, dev)
    log16, MN.LAUNCH_LOG = MN.LAUNCH_LOG, None
    full16 = [(B, a_.elapsed_time(b_)) for (flags, B, a_, b_) in log16 if (flags & 0x1000) and not (flags & 1)]
    sig16 = [(B, a_.elapsed_time(b_)) for (flags, B, a_, b_) in log16 if (flags & 0x1000) and (flags & 1)]
    sig32 = [(B, a_.elapsed_time(b_)) for (flags, B, a_, b_) in log if (flags & 0x1000) and (flags & 1)]
    for m in models.values():
        m.mlp_f16 = False
    ttr, tsrc, tcommit = _traffic("mnrf::mf::tcnn_encode_kernel")
    params = [p for m in models.values() for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=5e-4, fused=True)      # (one multi-tensor launch; the default "foreach" form is ~10 passes over the two 49 MB tables)
    target = torch.rand(1024, 3, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)          # every rank draws its own batch (DistributedSampler semantics)
    D.attach_overlap(models.values())  # (no-op on one rank) the all-reduces go out from inside the backward pass
    ar_ms, touched = [], []

    def step(measure=False):
        idx = torch.randint(0, rays.shape[0], (1024,), device=dev, generator=gen)
        res = M.render_rays(models, emb, rays[idx], N_SAMPLES, False, 1, 1, N_IMPORTANCE, compute_normal=False)
        loss = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean() \
            + 0.1 * ((res["mirror_mask_fine"] - 0.5) ** 2).mean() + 1e-4 * res["surface_normal_fine"].pow(2).sum(-1).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if measure:      # (host reads: outside the timed loop)
            touched.append([int((m.encoder.embeddings.grad != 0).any(-1).sum()) for m in models.values()])
        if collective:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            D.allreduce_gradients(params, modules=list(models.values()))
            e1.record()
            ar_ms.append((e0, e1))
        opt.step()
    for _ in range(3):
        step()
    step(measure=True)
    torch.cuda.synchronize()
    del ar_ms[:]
    if collective:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt_t_own = (time.perf_counter() - t0) / 20
    dt_t = D.max_over_ranks(dt_t_own, dev)
    entries = int(models["fine"].encoder.embeddings.shape[0])
    D.detach_overlap(list(models.values()))
    allreduce = None
    if collective:
        t_all = torch.tensor([dt_own, dt_t_own], dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(t_all) for _ in range(world)]
        torch.distributed.all_gather(parts, t_all)
        per_rank = torch.stack(parts).cpu()
        allreduce = {"messages_per_step": 4, "bytes_per_step": int(sum(q.numel() for q in params) * 4),
                     "wait_ms_per_step_rank0": sum(a.elapsed_time(b) for a, b in ar_ms) / max(1, len(ar_ms)),
                     "frame_s_per_rank": [round(float(v), 4) for v in per_rank[:, 0]],
                     "train_ms_per_rank": [round(float(v) * 1e3, 3) for v in per_rank[:, 1]],
                     "imbalance_train": float(per_rank[:, 1].max() / per_rank[:, 1].min()),
                     "note": "per model: the table gradient (one tensor) all-reduced in place + one blob of the 11 MLP gradients, "
                             "issued from the post-accumulate hooks of the backward pass; wait = time spent inside "
                             "allreduce_gradients after the backward (what was not hidden behind it)"}
    sparse = {"touched_entries_per_step": touched[0], "table_entries": entries,
              "touched_fraction": [t / entries for t in touched[0]],
              "dense_bytes_per_model": entries * 8, "sparse_index_value_bytes_per_model": [t * 12 for t in touched[0]],
              "sparse_over_dense": [t * 12 / (entries * 8) for t in touched[0]],
              "note": "rows of the table (coarse, fine model) that a 1024-ray batch (65 536 + 196 608 samples x 128 corner reads) "
                      "touches.  A sparse index + float2 exchange (SURVEY 8e) moves 12 B per touched row PER RANK and needs an "
                      "all-gather of variable-size lists (world x that, then a local merge); the dense in-place all-reduce moves "
                      "8 B per row whatever the world size.  With half of the fine table touched by one rank's batch the sparse "
                      "form is not smaller at 1 rank and strictly larger from 2 ranks on: the dense all-reduce is kept"}
    # the same step with the table gradient of the big hashed levels accumulated in half2 by packed atomics (tinycudann's
    # gradient precision: models/mirror_nerf_tcnn.py:36-49 under train.py:586; module.table_grad_f16, off by default)
    for m in models.values():
        m.table_grad_f16 = True
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt_t16 = (time.perf_counter() - t0) / 10
    for m in models.values():
        m.table_grad_f16 = False
    n = rays.shape[0]
    return {"rays_per_s": world * n / dt_f, "samples_per_s": world * n * (2 * N_SAMPLES + N_IMPORTANCE) / dt_f, "frame_ms": dt_f * 1e3,
            "n_gpus": world, "scaling": "weak (every rank renders the frame / draws its own 1024-ray batch)",
            "train_ms_per_step": dt_t * 1e3, "train_rays_per_s": world * 1024 / dt_t, "allreduce": allreduce,
            "table_gradient_sparsity": sparse,
            "train_ms_per_step_f16_table_grads": dt_t16 * 1e3,
            "train_table_gradient": "default: one packed 64-bit fixed-point atomic per entry (MNRF_TCNN_GRAD_FIXED, exact integer sums); _f16_table_grads: packed half2 atomics",
            "gather_roofline": {"bound": "l2", "kernel": "mnrf::mf::tcnn_encode_kernel (level-major encoding: 16 levels x 8 corners per sample; "
                                          "the dominant launch of the two-launch field evaluation)",
                                "achieved": (l2_bytes / (enc_ms * 1e-3) / 1e9) if l2_bytes else None, "peak": 34500.0, "unit": "GB/s",
                                "frac": (l2_bytes / (enc_ms * 1e-3) / 1e9 / 34500.0) if l2_bytes else None,
                                "l2_read_bytes_per_launch": l2_bytes, "l2_hit_rate": (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]))
                                if pmc and "TCC_HIT_sum" in pmc else None,
                                "tcp_hit_rate": (1.0 - pmc["TCP_TCC_READ_REQ_sum"] / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"])
                                if pmc and "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc else None,
                                "encode_ms": enc_ms, "field_ms_same_samples": same_ms, "encode_share_of_field": enc_ms / same_ms if same_ms else None,
                                "one_launch_form_ms": one_launch_ms, "G_gathers_per_s": zf.numel() * 128 / (enc_ms * 1e-3) / 1e9,
                                "context": {"independent_random_8B_gathers_in_a_2MiB_window": ceil["l2_resident_8B"],
                                            "uniformly_random_over_the_49MB_table": {"8_byte": ceil["random_8B"], "4_byte": ceil["random_4B"]}},
                                "counters_source": "static profile: profiles/traffic.json \"pmc\" (scripts/pmc_tcnn.sh), per launch of one 32768-ray chunk",
                                "note": "every gather that misses the 32 KB vector L1 pulls a 128-byte line out of the L2 for 8 useful bytes; "
                                        "achieved = those lines (PMC: TCP_TCC_READ_REQ x 128 B per launch) over the live launch time, against "
                                        "the L2's ~34.5 TB/s (MI355X_MICROARCH.md).  Round 3's one-launch kernel walked all 16 levels per wave: "
                                        "54 % of its lines missed the L2 too and 20.7 GB per launch crossed the fabric at 7.5 TB/s "
                                        "(profiles/r04b_pmc_tcnn); level by level the L2 holds the level (hit rate above)."},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": ttr,
                         "traffic_source": (f"static profile: profiles/traffic.json, commit {tcommit or 'unrecorded'}, {tsrc}") if ttr else None,
                         "kernel": "mnrf::mf::tcnn_encode_kernel + mnrf::mf::tcnn_mfma_kernel<0,true> (full evaluation, fine pass)",
                         "avg_launch_ms": ms_full / max(1, len(full)), "launches": len(full), "bytes_per_sample": 1024 + 256,
                         "algorithmic_bytes_per_launch": CHUNK * (N_SAMPLES + N_IMPORTANCE) * (1024 + 256),
                         "note": "ALGORITHMIC bytes (128 float2 table reads + the 128-byte encoding planes written and read once, per "
                                 "sample) over the time of the two launches; `traffic` = what reached the fabric in the encoding launch "
                                 "(PMC).  The table is cache-resident: gather_roofline (L2) is the ceiling that binds, this fraction "
                                 "is kept for the contract"},
            "f16_mlp": {"rays_per_s": world * n / dt_f16, "frame_ms": dt_f16 * 1e3,
                        "full_launch_ms": sum(t for _, t in full16) / max(1, len(full16)),
                        "sigma_only_launch_ms": sum(t for _, t in sig16) / max(1, len(sig16)),
                        "sigma_only_launch_ms_default_arithmetic": sum(t for _, t in sig32) / max(1, len(sig32)),
                        "note": "module.mlp_f16 / MNRF_TCNN_F16: single-pass f16 MLPs on the matrix pipe (one MFMA per product, fp32 "
                                "accumulation) -- \"fp16 MLP on CDNA4 MFMA\" as BASELINE config 5 words it, the arithmetic of tinycudann "
                                "under precision=16 (train.py:586); ~1e-3 relative to the default (tests); sigma-only launches on "
                                "the matrix pipe as well"},
            "note": "MirrorNeRFTcnn pair, random init; primary rays only (a random-init mask head predicts no mirror); parity "
                    "downstream of the encoder pinned by fixtures G17, the encoder's interpolation unpinned (DESIGN.md 2.2); full evaluations: MLPs as hi/lo f16 tiles on the matrix pipe, "
                    "sigma-only launches: the same two launches since round 4 (fp32 VALU kernel below 32768 samples)"}


def trained_leg(dev):
    """Scene-dependent effects need trained weights: the pair of fixture G11 (tests/golden/g11_trained_weights.npz: trained on the
    analytic mirror scene of make_golden_trained.py through this package, 19.1 dB held-out) rendered at 800x800 from a view of
    that scene (the scene_views camera at angle 0.2), eval rules with the PREDICTED mirror mask -- plain frame and ray-fused
    maps-only frame, with mean power x time = energy per frame and the clock.  Not `value` (random-init weights stay the headline)."""
    import numpy as np
    import torch
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN, synthetic as SY
    from mirror_nerf_amd.telemetry import SmiSampler
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "g11_trained_weights.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    models = {}
    for name in ("coarse", "fine"):
        m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
        m.load_state_dict({k[len(name) + 2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "__")})
        models[name] = m.to(dev)
    emb = {"xyz": M.Embedding(10), "dir": M.Embedding(4)}
    a = 0.2
    pose = SY.look_at_pose(eye=(2.6 * np.sin(a), -2.6 * np.cos(a) + 0.2, 0.9 + 0.5), target=(0.1, 0.6, 0.6))
    rays = SY.device_rays(H, W, dev, pose=pose, camera_angle_x=0.9)
    out = {}
    for key, kw in (("plain", {}), ("maps_only_fused", {"maps_only": True})):
        f = lambda: M.batched_inference(models, emb, rays, N_SAMPLES, N_IMPORTANCE, False, CHUNK, args=ARGS,  # noqa: E731
                                        trace_secondary_rays=True, to_cpu=False, **kw)
        r = f()
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        smi = SmiSampler(dev.index or 0, 0.2)
        with smi:
            t0 = time.perf_counter()
            for _ in range(2):
                r = f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
        n_launch = len(MN.LAUNCH_LOG) / 2
        MN.LAUNCH_LOG = None
        tele = smi.summary()
        pw = (tele.get("power_w") or {}).get("median")
        mask = r["mirror_mask_fine"]
        n_refl = float((mask > 0.5).sum())
        out[key] = {"rays_per_s": (H * W + n_refl) / dt, "ms_per_frame": dt * 1e3, "field_launches_per_frame": n_launch,
                    "power_w_median": pw, "energy_j_per_frame": pw * dt if pw else None,
                    "sclk_mhz_median": (tele.get("sclk_mhz") or {}).get("median")}
        out["reflected_rays_per_frame"] = n_refl
        if key == "plain":
            w = r.get("weights_fine")
            out["fine_samples_with_zero_weight"] = float((w == 0).float().mean()) if w is not None else None
    out["note"] = ("G11 trained pair, 800x800 view of its analytic scene, eval rules (predicted mask: reflected rays = mirror pixels "
                   "of chunks that hold any); fine_samples_with_zero_weight = the share of samples whose heads exact head skipping "
                   "(models/rendering.py:190-213: w_i = 0) could leave out -- not built, DESIGN.md 8")
    return out


def roughness_leg(dev, models, emb):
    """BASELINE config 4 at the shape run.sh:185-208 runs it (mode 5, control_mirror_roughness): 480x360 (run.sh:47-48),
    64 coarse + 64 importance samples, chunk 16384, one bounce, trace_ray_times = 64 jittered reflections per mirror ray,
    normal_noise_std = 0.0025, every pixel a mirror (so that the reference's level-0 addition is well-formed):
    172 800 primary + 65 x 172 800 reflected rays per frame.  The 64 jittered renders of a chunk go through the recursion
    in groups (recursion.JITTER_RAYS) instead of one by one.  Also: two bounces with eval.py's default trace_ray_times = 4
    (the jitters nest: (1 + 5) + 5 x ... renders per level)."""
    import torch
    import mirror_nerf_amd as M
    from mirror_nerf_amd import synthetic as SY
    Hc, Wc = 360, 480
    rays = SY.device_rays(Hc, Wc, dev)
    out = {}
    for name, levels, times in (("one_bounce_64_jitters", 1, 64), ("two_bounces_4_jitters", 2, 4)):
        args = dict(ARGS, max_recursive_level=levels, app_control_mirror_roughness=True, trace_ray_times=times)

        def frame():
            return M.batched_inference(models, emb, rays, N_SAMPLES, 64, False, 16384, args=args, trace_secondary_rays=True,
                                       normal_noise_std=0.0025, to_cpu=False)
        if name.startswith("one"):
            M.batched_inference(models, emb, rays[:16384], N_SAMPLES, 64, False, 16384, args=args, trace_secondary_rays=True,
                                normal_noise_std=0.0025, to_cpu=False)      # warm-up on one chunk
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = Hc * Wc
        rays_total = n * (1 + (times + 1)) if levels == 1 else n * (1 + (times + 1) * (1 + (times + 1)))
        out[name] = {"rays_per_s": rays_total / dt, "frame_s": dt, "rays_per_frame": rays_total,
                     "samples_per_s": rays_total / dt * (2 * N_SAMPLES + 64)}
    out["note"] = ("eval.batched_inference with app_control_mirror_roughness on the all-mirror random-init pair, 480x360, "
                   "64+64 samples, chunk 16384, normal_noise_std 0.0025 (run.sh:185-208); jittered reflections batched per level")
    return out


def clustered_balance(dev, models, rays, rank, world, sync):
    """Load balance of the ray sharding when mirror pixels CLUSTER: one frame under train semantics
    (NeRFSystem.forward, train.py:102-348) with a ground-truth mirror mask = the centred rectangle covering 25 % of the
    pixels and only_trace_rays_in_mirrors -- each rank renders its interleaved 4096-ray tiles plus the reflections of the
    mirror pixels among them.  Reports every rank's frame time and reflected-ray count, and max / mean of both; next to it
    the reflected-ray counts a CONTIGUOUS stripe per rank would get (computed, not rendered): the reason for interleaving."""
    import torch
    import torch.distributed as dist
    from mirror_nerf_amd import dist as D
    from mirror_nerf_amd import training
    from mirror_nerf_amd.recursion import NeRFSystem
    hp = training.default_hparams(N_importance=N_IMPORTANCE, perturb=0.0, noise_std=0.0, chunk=CHUNK)
    system = NeRFSystem(hp).to(dev)
    system.nerf_coarse.load_state_dict(models["coarse"].state_dict())
    system.nerf_fine.load_state_dict(models["fine"].state_dict())
    gt = torch.zeros(H, W, device=dev)
    gt[H // 4: H - H // 4, W // 4: W - W // 4] = 1.0
    gt = gt.view(-1)
    idx = D.shard_indices(rays.shape[0], rank, world, D.TILE, rays.device)
    r, m = rays[idx].contiguous(), gt[idx].contiguous()
    extra = {"mirror_mask": m, "is_eval": False, "train_geometry_stage": False}
    with torch.no_grad():
        system(r[:CHUNK], {k: (v[:CHUNK] if torch.is_tensor(v) else v) for k, v in extra.items()})      # warm-up
        sync()
        t0 = time.perf_counter()
        system(r, extra)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    mine = torch.tensor([dt, float(m.sum().item()), float(idx.numel())], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    allr = torch.stack(allr).cpu()
    times, refl = allr[:, 0].tolist(), allr[:, 1].tolist()
    stripe = rays.shape[0] // world
    contiguous = [float(gt[k * stripe: (k + 1) * stripe].sum().item()) for k in range(world)]

    def imb(v):
        mean = sum(v) / len(v)
        return max(v) / mean if mean > 0 else None
    return {"frame_s_per_rank": times, "reflected_rays_per_rank": refl, "primary_rays_per_rank": allr[:, 2].tolist(),
            "time_imbalance_max_over_mean": imb(times), "reflected_imbalance_max_over_mean": imb(refl),
            "contiguous_stripes_reflected_per_rank": contiguous, "contiguous_stripes_imbalance_max_over_mean": imb(contiguous),
            "note": "train semantics, GT mirror mask = centred 25 % rectangle, only_trace_rays_in_mirrors; interleaved 4096-ray tiles "
                    "(rendered) vs contiguous stripes (counted only)"}


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


def _pmc(kernel):
    """Counters of `kernel` (one launch of a 32768-ray chunk) from the static profile, {} when absent."""
    try:
        with open(TRAFFIC_JSON) as f:
            return json.load(f).get("pmc", {}).get(kernel, {})
    except (OSError, ValueError):
        return {}


def _traffic(kernel):
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        e = t.get(kernel)
        return (e["hbm_bytes_per_launch"], e.get("source"), e.get("commit")) if e else (None, None, None)
    except (OSError, ValueError, KeyError):
        return None, None, None


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
    strong = other_tf = other_rays = other_ms = smi_other = host_maps = fused = train = hash_grid = rough = trained = None
    legs_done = []

    def emit(incomplete=None):
        """Rank 0 prints the ONE line.  `incomplete`: legs cut short by the deadline below (N > 1 only)."""
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
                           "collective_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (MNRF_SHARE_GPU test aid: timings void)")) if multi else None,
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
            if strong is not None:
                res["strong_scaling"] = strong
            if host_maps is not None:
                res["with_host_maps"] = host_maps
            if fused is not None:
                res["maps_only_fused"] = fused
            if train is not None:
                res["train_step"] = train
            if hash_grid is not None:
                res["hash_grid_variant"] = hash_grid
            if rough is not None:
                res["roughness_variant"] = rough
            if trained is not None:
                res["trained_weights_variant"] = trained
            if world == 1 and not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(sds, a.cpu_seconds)
            if incomplete:
                res["incomplete_legs"] = incomplete
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
                os._exit(0)
        deadline = threading.Timer(float(os.environ.get("MNRF_BENCH_LEG_DEADLINE", "900")), _expired)
        deadline.daemon = True
        deadline.start()

    # strong scaling: ONE frame (rank 0's view) dealt to the ranks in interleaved 4096-ray tiles; no collective while
    # rendering; then the optional assembly of the 20 B/ray maps on rank 0
    strong = None
    if multi:
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
    if world == 1:
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
    if world == 1:
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
    if not a.no_train:
        from mirror_nerf_amd import training
        train = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024)
        legs_done.append("train_step")
        train_total = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, loss_name="total")
        train["with_total_loss"] = {k: train_total[k] for k in ("value", "ms_per_step", "loss", "loss_fn", "roofline")
                                    if k in train_total}
        # BASELINE config 3 as worded ("same config" as config 2: 64 coarse + 128 importance samples); the default above is
        # run.sh:266's training schedule (--N_importance 64)
        train_c3 = training.synthetic_train_bench(dev, rays, steps=30, warmup=5, batch=1024, N_importance=128)
        train["config3_64_plus_192"] = {k: train_c3[k] for k in ("value", "ms_per_step", "samples_per_ray", "N_importance", "roofline",
                                                                  "reflected_rays_per_step", "allreduce")}
        legs_done.append("train_step.with_total_loss + config3_64_plus_192")

    hash_grid = hash_grid_leg(dev, rays) if not a.no_train else None
    legs_done.append("hash_grid_variant")
    rough = roughness_leg(dev, models, emb) if (world == 1 and not a.no_train) else None
    trained = trained_leg(dev) if (world == 1 and not a.no_train) else None

    if deadline is not None:
        deadline.cancel()
    emit()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
