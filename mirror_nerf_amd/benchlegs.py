"""Secondary legs of bench.py (round 5: split out so that bench.py's timed region reads in one screen).  Nothing here is the
headline measurement: `value` / `roofline` are taken in bench.main() before any of these runs.  GPU only; the CPU baseline stays in
bench.py (it is the one place outside tests/ that may call into oracle/).

    hash_grid_leg      BASELINE config 5 (hash-grid field): frame, training step, gather roofline, all-reduce
    trained_leg        the G11 trained pair at 800x800: rays/s, energy per frame, zero-weight share
    roughness_leg      BASELINE config 4 (roughness jitters; run.sh:185-208)
    config1_leg        BASELINE config 1 (400x400, coarse-only 64 samples, train semantics) on the GPU
    clustered_balance  load balance of the ray sharding with a clustered mirror mask (N > 1)
"""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W = 800
N_SAMPLES, N_IMPORTANCE, CHUNK = 64, 128, 32768
ARGS = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)
# roofline.traffic: HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes of THIS command
# (scripts/pmc_passes.sh -> profiles/traffic.json, keyed by kernel name; FETCH_SIZE doubled per the guide's gfx950
# correction + WRITE_SIZE).  PMC counters cannot be read from inside the run: a kernel without an entry reports null.
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


def _pmc(kernel):
    """Counters of `kernel` (one launch of a 32768-ray chunk) from the static profile, {} when absent."""
    try:
        with open(TRAFFIC_JSON) as f:
            return json.load(f).get("pmc", {}).get(kernel, {})
    except (OSError, ValueError):
        return {}


def _l1_rate(pmc, ms):
    """Vector-L1 (TCP) line accesses of a launch per clock and CU at the nominal 2.4 GHz.  Reported as context for the level-major
    gather launch (round 5, 32-ray x 8-depth patches per workgroup: most gathers hit the L1 and the L2 line rate is no longer what
    binds); the guide gives no peak for it and an experiment that removed a quarter of these accesses (the two x-neighbours of a
    corner as one 16-byte gather where they are adjacent, bit-identical planes) left the time unchanged, so it is not the bound either."""
    if not pmc or "TCP_TOTAL_CACHE_ACCESSES_sum" not in pmc or not ms:
        return None
    return {"l1_line_accesses_per_launch": pmc["TCP_TOTAL_CACHE_ACCESSES_sum"],
            "per_clock_per_cu_at_2400MHz": pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * ms * 2.4e6)}


def _traffic(kernel):
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        e = t.get(kernel)
        return (e["hbm_bytes_per_launch"], e.get("source"), e.get("commit")) if e else (None, None, None)
    except (OSError, ValueError, KeyError):
        return None, None, None


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
    # round 5: the table in half2 -- 4 B per entry, tinycudann's storage (models/mirror_nerf_tcnn.py:39-49; SURVEY 8d prices config 5
    # at 16 x 8 x 4 = 512 B of gathers per sample) -- with the default MLP arithmetic and with single-pass f16 MLPs (config 5 as worded)
    f16_table = {}
    for m in models.values():
        m.table_f16 = True
    for key, mlp16 in (("frame_ms", False), ("frame_ms_f16_mlp", True)):
        for m in models.values():
            m.mlp_f16 = mlp16
        frame()
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize()
        f16_table[key] = D.max_over_ranks(time.perf_counter() - t0, dev) * 1e3
        log_w, MN.LAUNCH_LOG = MN.LAUNCH_LOG, None
        if mlp16:
            full_w = [(B, a_.elapsed_time(b_)) for (flags, B, a_, b_) in log_w if (flags & 0x1000) and not (flags & 1)]
    th, tflag = mfine._table()
    planes = torch.empty(32 * zf.numel(), device=dev)
    pr16 = lambda: _lib.check(_lib.lib().mnrf_tcnn_encode_flags(  # noqa: E731
        th.data_ptr(), offs, mfine.cfg["S"], mfine.cfg["H"], float(mfine.bound), zf.numel(), None, 0,
        _lib.ptr(rchunk), _lib.ptr(zf), zf.shape[1], _lib.ptr(planes), tflag, _lib.stream()), "encode (half2 table)")
    pr16()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pr16()
    e1.record()
    torch.cuda.synchronize()
    enc16_ms = e0.elapsed_time(e1) / 5
    planes = None
    for m in models.values():
        m.table_f16 = m.mlp_f16 = False
    pmc16 = _pmc("mnrf::mf::tcnn_encode_kernel [half2 table]")
    l2_16 = pmc16.get("TCP_TCC_READ_REQ_sum", 0.0) * 128.0 if pmc16 else None
    n_ = rays.shape[0]
    f16_table.update({
        "rays_per_s": world * n_ / (f16_table["frame_ms"] * 1e-3), "rays_per_s_f16_mlp": world * n_ / (f16_table["frame_ms_f16_mlp"] * 1e-3),
        "encode_ms": enc16_ms, "encode_ms_fp32_table": enc_ms, "table_bytes": int(th.numel() * 2),
        "gather_roofline": {"bound": "l2", "achieved": (l2_16 / (enc16_ms * 1e-3) / 1e9) if l2_16 else None, "peak": 34500.0, "unit": "GB/s",
                            "frac": (l2_16 / (enc16_ms * 1e-3) / 1e9 / 34500.0) if l2_16 else None, "l2_read_bytes_per_launch": l2_16,
                            "l2_hit_rate": (pmc16["TCC_HIT_sum"] / (pmc16["TCC_HIT_sum"] + pmc16["TCC_MISS_sum"])) if pmc16 and "TCC_HIT_sum" in pmc16 else None,
                            "tcp_hit_rate": (1.0 - pmc16["TCP_TCC_READ_REQ_sum"] / pmc16["TCP_TOTAL_CACHE_ACCESSES_sum"])
                            if pmc16 and "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc16 else None,
                            "hbm_bytes_per_launch": ((2 * pmc16["FETCH_SIZE"] + pmc16["WRITE_SIZE"]) * 1024) if pmc16 and "FETCH_SIZE" in pmc16 and "WRITE_SIZE" in pmc16 else None,
                            "l1_access_rate": _l1_rate(pmc16, enc16_ms),
                            "counters_source": pmc16.get("source") if pmc16 else None},
        "algorithmic_bytes_per_sample": 512 + 256,
        "note": "module.table_f16 / MNRF_TCNN_TABLE_F16: the kernels gather from a half2 copy of the table (2 MB per hashed level against "
                "the 4 MB L2 of an XCD; the fp32 master stays what the optimizer steps); forward bit-identical to the fp32-table kernels "
                "on a table rounded to f16 (tests).  With 32-ray x 8-depth patches per workgroup most gathers hit the vector L1 and the "
                "launch takes the same time with either entry size (`l1_access_rate`: 0.8-0.9 L1 line accesses per clock and CU): the half2 "
                "table halves the footprint and the fabric traffic, not the time"})
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
    recursion = _tcnn_recursion_step(dev, rays) if world == 1 else None
    return {"rays_per_s": world * n / dt_f, "samples_per_s": world * n * (2 * N_SAMPLES + N_IMPORTANCE) / dt_f, "frame_ms": dt_f * 1e3,
            "train_step_with_reflections": recursion,
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
                                "l1_access_rate": _l1_rate(pmc, enc_ms),
                                "counters_source": "static profile: profiles/traffic.json \"pmc\" (scripts/pmc_tcnn.sh), per launch of one 32768-ray chunk",
                                "note": "every gather that misses the 32 KB vector L1 pulls a 128-byte line out of the L2 for 8 useful bytes; "
                                        "achieved = those lines (PMC: TCP_TCC_READ_REQ x 128 B per launch) over the live launch time, against "
                                        "the L2's ~34.5 TB/s (MI355X_MICROARCH.md).  Round 3's one-launch kernel walked all 16 levels per wave: "
                                        "54 % of its lines missed the L2 too and 20.7 GB per launch crossed the fabric at 7.5 TB/s "
                                        "(profiles/r04b_pmc_tcnn); round 4 went level by level (L2 hit rate 0.95, 0.64 of this roofline); "
                                        "round 5 gives a workgroup a patch of 32 neighbouring rays x 8 depths: the L1 hit rate rises "
                                        "0.57 -> 0.83 and the lines asked of the L2 fall 2.5x -- this L2 fraction is lower than round 4's "
                                        "because the kernel needs less of the L2, not because it got slower (1.62 -> 1.24 ms).  What binds "
                                        "now is inside the CU's gather path (`l1_access_rate`: 0.9 L1 line accesses per clock and CU; "
                                        "removing a quarter of them did not shorten the launch, DESIGN 9)."},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": ttr,
                         "traffic_source": (f"static profile: profiles/traffic.json, commit {tcommit or 'unrecorded'}, {tsrc}") if ttr else None,
                         "kernel": "mnrf::mf::tcnn_encode_kernel + mnrf::mf::tcnn_mfma_kernel<0,true> (full evaluation, fine pass)",
                         "avg_launch_ms": ms_full / max(1, len(full)), "launches": len(full), "bytes_per_sample": 1024 + 256,
                         "survey_8d": _survey_roofline(full),
                         "algorithmic_bytes_per_launch": CHUNK * (N_SAMPLES + N_IMPORTANCE) * (1024 + 256),
                         "note": "ALGORITHMIC bytes (128 float2 table reads + the 128-byte encoding planes written and read once, per "
                                 "sample) over the time of the two launches; `traffic` = what reached the fabric in the encoding launch "
                                 "(PMC).  The table is cache-resident: gather_roofline (L2) is the ceiling that binds, this fraction "
                                 "is kept for the contract"},
            "f16_table": f16_table,
            # the configuration BASELINE words ("hash-grid encoding variant ... with fp16 MLP"), on the storage SURVEY
            # 8(d) prices (tinycudann's half2 entries): a named key with its own roofline on 8(d)'s algorithmic bytes
            "as_baseline_words_it": {
                "table": "half2 entries (module.table_f16; 4 B, tinycudann's storage)", "mlp": "single-pass f16 on the matrix pipe (module.mlp_f16)",
                "rays_per_s": f16_table.get("rays_per_s_f16_mlp"), "frame_ms": f16_table.get("frame_ms_f16_mlp"),
                "full_launch_ms": sum(t for _, t in full_w) / max(1, len(full_w)),
                "roofline": _survey_roofline(full_w),
                "parity": "fixture G17 (the reference's render over stand-in encoders) at 5e-3 of each output's scale: "
                          "tests/test_hip_tcnn.py::test_g17_render_rays_with_f16_table_and_f16_mlps; the interpolation itself is unpinned "
                          "(tinycudann absent)"},
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


def _tcnn_recursion_step(dev, rays, steps=20):
    """Config 5 through the WHOLE training path of the reference (train.py:102-348 with MODEL_TYPE=nerf_tcnn: coarse + fine pass, 25 % GT
    mirror rays reflected once, blend), colour + mask loss, torch's fused Adam: train_step on the host-driven route (one device->host
    read of the reflected-ray count per level and step) against the static route (round 6: mnrf_tcnn_forward_n / _backward_n)."""
    import torch
    import mirror_nerf_amd as M
    from mirror_nerf_amd import training as T
    out = {}
    for route in ("host", "static"):
        torch.manual_seed(0)
        hp = T.default_hparams(model_type="nerf_tcnn", bound=6.0, N_emb_xyz=0, N_emb_dir=0, perturb=1.0, noise_std=1.0)
        system = M.NeRFSystem(hp).to(dev)
        with torch.no_grad():
            for m in system.models.values():
                m.sigma_net[1].weight[0] *= 10.0       # opaque enough for surfaces (and reflected rays) to exist
        opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=True)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        batches = []
        for _ in range(steps + 3):
            idx = torch.randint(0, rays.shape[0], (1024,), device=dev, generator=g)
            batches.append((rays[idx].contiguous(), torch.rand(1024, 3, device=dev, generator=g),
                            (torch.rand(1024, device=dev, generator=g) < 0.25).float()))
        it = iter(batches)
        for _ in range(3):
            T.train_step(system, opt, *next(it), gt_valid=True if route == "static" else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = T.train_step(system, opt, *next(it), gt_valid=True if route == "static" else None)
        torch.cuda.synchronize()
        out[route + "_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
        out["loss_" + route] = float(loss.detach())
        del system, opt
    out["note"] = ("NeRFSystem(model_type='nerf_tcnn').forward with reflections (1024 rays, 64 + 128 samples, ~256 reflected rays) + colour / "
                   "mask loss + backward + fused Adam through training.train_step: host-driven route vs static route (the reflected-ray count "
                   "stays on the device; the hash-grid kernels take it as their live row count)")
    return out


def _survey_roofline(launches):
    """SURVEY 8(d)'s algorithmic bytes of a hash-grid sample: 16 levels x 8 corners x 4 B (half2 entries) = 512 B of gathers + 24 B
    in (position, direction) = 536 B, over the time of the full-evaluation launches (encoding + MLP), against the HBM peak."""
    ms = sum(t for _, t in launches)
    if not launches or ms <= 0:
        return None
    gbs = sum(B for B, _ in launches) * 536 / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "bytes_per_sample": 536, "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
            "note": "SURVEY 8(d): 512 B of table gathers (half2) + 24 B in per sample; the table is cache-resident, so this is the "
                    "contract's yardstick, not the binding limit (gather_roofline: L2 line fills)"}


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
    path = os.path.join(ROOT, "tests", "golden", "g11_trained_weights.npz")
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
        # rays the launches PROCESSED: under eval rules level 0 traces every ray of a chunk that holds a mirror pixel (eval.py:159),
        # not only the mirror pixels -- full (4-head) launches evaluate N_SAMPLES + N_IMPORTANCE samples per ray
        rendered = sum(B for (flags, B, _e0, _e1) in MN.LAUNCH_LOG if not (flags & 1)) / (N_SAMPLES + N_IMPORTANCE) / 2
        MN.LAUNCH_LOG = None
        tele = smi.summary()
        pw = (tele.get("power_w") or {}).get("median")
        mask = r["mirror_mask_fine"]
        n_refl = float((mask > 0.5).sum())
        out[key] = {"rays_per_s": (H * W + n_refl) / dt, "rays_per_s_counts": "credited: H x W + mirror pixels (what the frame shows)",
                    "rendered_rays_per_frame": rendered, "rendered_rays_per_s": rendered / dt,
                    "rendered_note": "rays the field launches evaluated: every ray of a chunk that holds a mirror pixel is traced at "
                                     "level 0 (eval.py:159); the kernels run at the headline's rate on THIS count",
                    "ms_per_frame": dt * 1e3, "field_launches_per_frame": n_launch,
                    "power_w_median": pw, "energy_j_per_frame": pw * dt if pw else None,
                    "sclk_mhz_median": (tele.get("sclk_mhz") or {}).get("median")}
        out["reflected_rays_per_frame"] = n_refl
        if key == "plain":
            w = r.get("weights_fine")
            out["fine_samples_with_zero_weight"] = float((w == 0).float().mean()) if w is not None else None
    out["guard_trips"] = int(sum(m.__dict__.get("_mnrf_guard_trips", 0) for m in models.values()))      # range-guard trips of the frames above
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
    # config 4 exactly as run.sh:185-208 words it -- 2 bounces AND trace_ray_times = 64 -- nests the jitters: every mirror ray of
    # level 0 spawns 65 renders at level 1, each of which spawns 65 at level 2 (1 + 65 + 65^2 = 4291 rays per pixel).  A
    # 480x360 frame of that is 741 M rays; measured on a reduced 48x36 frame (all-mirror)
    Hs, Ws = 36, 48
    small = SY.device_rays(Hs, Ws, dev)
    args = dict(ARGS, max_recursive_level=2, app_control_mirror_roughness=True, trace_ray_times=64)
    f2 = lambda: M.batched_inference(models, emb, small, N_SAMPLES, 64, False, 16384, args=args, trace_secondary_rays=True,  # noqa: E731
                                     normal_noise_std=0.0025, to_cpu=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f2()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = Hs * Ws
    rays_total = n * (1 + 65 * (1 + 65))
    out["two_bounces_64_jitters"] = {"rays_per_s": rays_total / dt, "frame_s": dt, "rays_per_frame": rays_total, "frame": f"{Ws}x{Hs}",
                                     "samples_per_s": rays_total / dt * (2 * N_SAMPLES + 64),
                                     "note": "config 4 as run.sh:185-208 words it (2 bounces AND 64 jittered reflections per mirror ray, "
                                             "which nest: 4291 rays per pixel) on a reduced all-mirror frame"}
    out["note"] = ("eval.batched_inference with app_control_mirror_roughness on the all-mirror random-init pair, 480x360, "
                   "64+64 samples, chunk 16384, normal_noise_std 0.0025 (run.sh:185-208); jittered reflections batched per level")
    return out


def config1_leg(dev):
    """BASELINE config 1 on the GPU, beside the CPU plumbing path it is worded for (bench.config1_cpu): the 400x400 synthetic
    camera, coarse-only 64 samples (N_importance = 0), MODEL_TYPE=nerf, one reflection bounce under TRAIN semantics
    (NeRFSystem.forward, train.py:102-348) with the ground-truth mirror mask of SURVEY 8d (centred rectangle, 25 % of the
    pixels, only_trace_rays_in_mirrors): 160 000 primary + 40 000 reflected rays per frame, chunk 32768.  Parity of this
    configuration: fixture G15."""
    import torch
    from mirror_nerf_amd import synthetic as SY, training
    from mirror_nerf_amd.recursion import NeRFSystem
    Hc = Wc = 400
    torch.manual_seed(0)
    hp = training.default_hparams(N_importance=0, perturb=0.0, noise_std=0.0, chunk=CHUNK)
    system = NeRFSystem(hp).to(dev)
    with torch.no_grad():
        system.nerf_coarse.sigma.weight.mul_(20.0)      # opaque density: surfaces (and reflections off them) exist
        system.nerf_coarse.sigma.bias.fill_(1.0)
    rays = SY.device_rays(Hc, Wc, dev)
    gt = torch.zeros(Hc, Wc, device=dev)
    gt[Hc // 4: Hc - Hc // 4, Wc // 4: Wc - Wc // 4] = 1.0
    extra = {"mirror_mask": gt.view(-1), "is_eval": False, "train_geometry_stage": False}
    with torch.no_grad():
        system(rays, extra)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = system(rays, extra)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    n_refl = float(gt.sum().item())
    return {"rays_per_s": (Hc * Wc + n_refl) / dt, "ms_per_frame": dt * 1e3, "primary_rays": Hc * Wc, "reflected_rays": n_refl,
            "samples_per_s": (Hc * Wc + n_refl) * N_SAMPLES / dt, "keys": sorted(k for k in out if k.startswith("rgb")),
            "note": "400x400, coarse-only 64 full samples per ray (density-gradient normal on: compute_normal = trace_secondary_rays, "
                    "train.py:143), GT mirror rectangle 25 %, reflected rays compacted, one bounce; forward only (the training "
                    "step is train_step); random-init coarse model with opaque density"}


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


