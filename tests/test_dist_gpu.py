"""RCCL on the GPU box: the collectives of mirror_nerf_amd.dist through backend "nccl" (= RCCL on ROCm).

A 1-GPU box can only form a 1-rank group; MNRF_FORCE_COLLECTIVES=1 makes dist.py issue the collectives anyway, so the
code path of the 8-GPU run (init_from_env -> allreduce_gradients / max_over_ranks / gather_frame on device tensors, the
sharded render through the HIP path) executes here.  The 2-rank variant runs wherever two GPUs are visible.
Each case runs in child processes (a process group cannot be re-initialised inside the pytest process)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["MNRF_ROOT"])
    import torch, torch.distributed as dist
    import mirror_nerf_amd as M
    from mirror_nerf_amd import dist as D, synthetic as SY, training

    rank, ws, dev = D.init_from_env()
    share = os.environ.get("MNRF_SHARE_GPU") == "1"      # (tests/shared_gpu/sitecustomize.py is loaded then)
    assert dist.is_initialized() and dist.get_backend() == ("gloo" if share else "nccl") and dist.get_world_size() == ws
    assert dev.index == (0 if share else int(os.environ["LOCAL_RANK"])) == torch.cuda.current_device()

    # max over ranks / flat gradient all-reduce on device tensors
    assert D.max_over_ranks(1.0 + rank, dev) == float(ws)
    p1 = torch.nn.Parameter(torch.zeros(5, device=dev)); p1.grad = torch.full((5,), float(rank + 1), device=dev)
    p2 = torch.nn.Parameter(torch.zeros(2, 3, device=dev)); p2.grad = torch.full((2, 3), 10.0 * (rank + 1), device=dev)
    D.allreduce_gradients([p1, p2])
    mean = sum(range(1, ws + 1)) / ws
    assert torch.allclose(p1.grad, torch.full((5,), mean, device=dev)) and torch.allclose(p2.grad, torch.full((2, 3), 10 * mean, device=dev))

    # a frame dealt to the ranks in interleaved tiles through the HIP path, assembled on rank 0, equals the unsharded render
    models, _ = SY.build_models(dev, SY.STRADDLE, seed=0)
    emb = {"xyz": M.Embedding(10), "dir": M.Embedding(4)}
    rays = SY.device_rays(24, 24, dev)
    args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)
    render = lambda r: M.batched_inference(models, emb, r, 64, 64, False, 32768, args=args, trace_secondary_rays=True, to_cpu=False)
    keys = ("rgb_fine", "depth_fine", "mirror_mask_fine")
    idx, res = D.render_sharded(render, rays, tile=100)
    full = D.gather_frame(idx, res, rays.shape[0], keys=keys, tile=100)
    if rank == 0:
        want = render(rays)
        for k in keys:
            assert torch.equal(full[k], want[k]), k

    # the training step with the flat all-reduce: every rank ends with identical parameters
    r = training.synthetic_train_bench(dev, SY.device_rays(64, 64, dev), steps=2, warmup=1, batch=256)
    assert r["ms_per_step"] > 0
    # both models reduced through their flat gradient buffer in place, the all-reduce issued from inside the backward pass
    assert r["allreduce"] == dict(r["allreduce"], buckets_in_place=2, models=2, overlapped_with_backward=2), r["allreduce"]
    # hash-grid models (BASELINE config 5 "... 8xMI355X"): the table gradient is all-reduced in place from inside the backward
    # pass, the MLP gradients as one blob; every rank ends the step with the same averaged gradients
    torch.manual_seed(0)
    tm = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev) for k in ("coarse", "fine")}
    assert not D._SEQ                # synthetic_train_bench detached its system's buckets (buckets go out in a fixed order)
    ov = D.attach_overlap(tm.values())
    assert len(ov) == 2
    emb0 = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    g = torch.Generator(device=dev); g.manual_seed(1 + rank)
    rr = SY.device_rays(32, 32, dev)
    res = M.render_rays(tm, emb0, rr[torch.randint(0, rr.shape[0], (256,), device=dev, generator=g)], 32, False, 1, 1, 32, compute_normal=False)
    (res["rgb_fine"].pow(2).mean() + res["rgb_coarse"].pow(2).mean() + 0.1 * res["mirror_mask_fine"].mean()).backward()
    assert all(o.work is not None and len(o.work) == 2 for o in ov)
    local = {k: m.encoder.embeddings.grad for k, m in tm.items()}
    D.allreduce_gradients([q for m in tm.values() for q in m.parameters()], modules=list(tm.values()))
    for k, m in tm.items():
        assert m.encoder.embeddings.grad is local[k]                 # reduced in place: no cat, no copy back
        chk = torch.stack([m.encoder.embeddings.grad.double().sum(), m.sigma_net[0].weight.grad.double().sum()])
        parts = [torch.zeros_like(chk) for _ in range(ws)]
        dist.all_gather(parts, chk)
        assert all(torch.equal(parts[0], q) for q in parts) and float(chk[0].abs()) > 0
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank)
''')


GUARD_WORKER = textwrap.dedent('''
    import os, sys, warnings
    sys.path.insert(0, os.environ["MNRF_ROOT"])
    import torch, torch.distributed as dist
    import mirror_nerf_amd as M
    from mirror_nerf_amd import dist as D, synthetic as SY, training
    from mirror_nerf_amd import mirror_nerf as MN

    rank, ws, dev = D.init_from_env()
    mode = training.GUARD_MODE
    torch.manual_seed(0)
    system = M.NeRFSystem(training.default_hparams()).to(dev)          # same weights on every rank
    if os.environ.get("MNRF_TEST_OPT") == "flat":      # the bench's optimizer: Adam over one flat tensor per model (mnrf_adam_step)
        opt = training.FlatAdam(list(system.models.values()), lr=5e-4)
    else:
        opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=True)
    D.attach_overlap(system.models.values())
    g = torch.Generator(device=dev); g.manual_seed(1 + rank)
    all_rays = SY.device_rays(64, 64, dev)

    def batch(far_away):
        rays = all_rays[torch.randint(0, all_rays.shape[0], (256,), device=dev, generator=g)].contiguous()
        if far_away:
            rays[:, :3] += 100.0          # positions beyond the fast sin/cos range: the encoding-range bit of the guard word
        return rays, torch.rand(256, 3, device=dev, generator=g), (torch.rand(256, device=dev, generator=g) < 0.25).float()

    def checksum():
        c = torch.stack([q.detach().double().sum() for q in system.parameters()]).sum().reshape(1)
        parts = [torch.zeros_like(c) for _ in range(ws)]
        dist.all_gather(parts, c)
        return [float(q) for q in parts]

    before = [q.detach().clone() for q in system.parameters()]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        training.train_step(system, opt, *batch(far_away=(rank == ws - 1)))      # ONLY the last rank leaves the range
        torch.cuda.synchronize()
        same = all(torch.equal(a, q.detach()) for a, q in zip(before, system.parameters()))
        if mode == "skip":
            assert same, "skip: the update of the tripping step must be skipped on EVERY rank"
        else:
            assert not same, "sync: the step is recomputed on the exact kernels and applied on every rank"
            assert all(MN.precision_of(m) == "fp32" for m in system.models.values())
        c1 = checksum()
        training.train_step(system, opt, *batch(False))
        torch.cuda.synchronize()
    assert all(MN.precision_of(m) == "fp32" for m in system.models.values()), "every rank pins every model, whoever tripped"
    c2 = checksum()
    assert all(abs(c - c1[0]) <= 1e-9 * max(1.0, abs(c1[0])) for c in c1), c1      # the ranks hold the same weights
    assert all(abs(c - c2[0]) <= 1e-9 * max(1.0, abs(c2[0])) for c in c2), c2
    assert c2[0] != c1[0], ("the step after the trip did not update the weights", c1, c2)
    assert all(bool(torch.isfinite(q).all()) for q in system.parameters())
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank)
''')


GRAPH_WORKER = textwrap.dedent('''
    import os, sys, warnings
    sys.path.insert(0, os.environ["MNRF_ROOT"])
    import torch, torch.distributed as dist
    import mirror_nerf_amd as M
    from mirror_nerf_amd import dist as D, synthetic as SY, training as T
    from mirror_nerf_amd.weights import params_of

    rank, ws, dev = D.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and (ws > 1 or D.forced())

    def system():
        torch.manual_seed(0)
        s = M.NeRFSystem(T.default_hparams(perturb=0.0, noise_std=0.0)).to(dev)
        with torch.no_grad():
            for m in s.models.values():
                m.sigma.weight.mul_(20.0); m.sigma.bias.fill_(1.0)
        return s

    g = torch.Generator(device=dev); g.manual_seed(1 + rank)
    all_rays = SY.device_rays(64, 64, dev)
    n = 256
    batches = []
    for i in range(4):
        rays = all_rays[torch.randint(0, all_rays.shape[0], (n,), device=dev, generator=g)].contiguous()
        batches.append((rays, torch.rand(n, 3, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < (0.0 if i == 1 else 0.3)).float()))

    calls = []
    real = dist.all_reduce
    def counted(t, *a, **k):
        calls.append((int(t.numel()), str(t.dtype), bool(torch.cuda.is_current_stream_capturing())))
        return real(t, *a, **k)
    dist.all_reduce = counted

    # -- the step captured WITH its collectives: two bucket all-reduces from the backward hooks + the OR of the guard words
    a = system()
    opt_a = T.FlatAdam(list(a.models.values()), lr=5e-4)
    D.attach_overlap(a.models.values())
    step = T.GraphedTrainStep(a, opt_a, n, gt_valid=True)
    assert step.collective
    snaps_a, losses_a, grads_a = [], [], None
    for i, b in enumerate(batches):
        la = step(*b)
        torch.cuda.synchronize()
        assert not step.ended and step.capture_error is None, step.capture_error
        losses_a.append(float(la))
        if i == 0:
            grads_a = [D._flat_bucket(m).clone() for m in a.models.values()]
        snaps_a.append([q.detach().clone() for m in a.models.values() for q in params_of(m)])
    captured = [c for c in calls if c[2]]
    n_flat = sum(q.numel() for q in params_of(a.nerf_fine))
    assert sorted(c[0] for c in captured) == sorted([n_flat, n_flat, 2 * len(T.GraphedTrainStep._BITS)]), captured
    n_calls = len(calls)
    D.detach_overlap(list(a.models.values()))

    # -- the same batches through train_step's static route (host-issued launches, host-issued collectives)
    b_sys = system()
    opt_b = T.FlatAdam(list(b_sys.models.values()), lr=5e-4)
    D.attach_overlap(b_sys.models.values())
    for i, b in enumerate(batches):
        lb = T.train_step(b_sys, opt_b, *b, gt_valid=True)
        torch.cuda.synchronize()
        if i == 0:
            assert float(lb) == losses_a[0], (float(lb), losses_a[0])
            for ga, m in zip(grads_a, b_sys.models.values()):
                assert torch.equal(ga, D._flat_bucket(m)), "gradients of the replayed step differ from train_step's"
        assert abs(float(lb) - losses_a[i]) <= 1e-6 * max(1.0, abs(float(lb))), (i, float(lb), losses_a[i])
        worst = max(float((qa - qb.detach()).abs().max()) for qa, qb in zip(snaps_a[i], (q for m in b_sys.models.values() for q in params_of(m))))
        assert worst <= 2.5e-4 * (i + 1), (i, worst)      # (one ulp in a gradient near zero can flip an Adam step of lr)
        print("step", i, "loss", float(lb), "max |w_graph - w_static|", worst)
    assert len(calls) > n_calls          # (train_step issued its collectives from the host)
    D.detach_overlap(list(b_sys.models.values()))
    dist.all_reduce = real
    # replays issue no Python-side collective: the 4 calls of the graphed run made 3 (capture) + 3 per warm-up rehearsal
    assert n_calls == 3 * (1 + step.warmup), (n_calls, calls[:12])
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank)
''')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SHIM = os.path.join(ROOT, "tests", "shared_gpu")      # sitecustomize.py: the MNRF_SHARE_GPU transport shim (test aid, not product code)


def _share_env(env):
    """MNRF_SHARE_GPU=1 only means something to processes that load tests/shared_gpu/sitecustomize.py."""
    if env.get("MNRF_SHARE_GPU") == "1":
        env["PYTHONPATH"] = SHIM + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        env["MNRF_ROOT"] = ROOT
    return env


def _run(world, worker=None, **extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, **extra, MNRF_ROOT=ROOT, MNRF_FORCE_COLLECTIVES="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env = _share_env(env)
        procs.append(subprocess.Popen([sys.executable, "-c", worker or WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append((p.returncode, out))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"RANK_OK {r}" in out, out[-3000:]


def test_rccl_one_rank_group_runs_every_collective():
    _run(1)


def test_graphed_step_with_forced_collectives():
    """the bucket all-reduces and the guard-word reduction captured INSIDE the hipGraph of the training step
    (RCCL, world size 1 with MNRF_FORCE_COLLECTIVES=1 -- the group a 1-GPU box can form).  Replays issue no collective from Python;
    loss and gradients of the first replay are bit-identical to train_step's static route on the same batch, weights after four
    steps to one Adam step's arithmetic.  DDP semantics: train.py:577-584."""
    _run(1, GRAPH_WORKER)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_graphed_step_with_collectives_two_ranks():
    _run(2, GRAPH_WORKER)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_rccl_two_ranks():
    _run(2)


def test_two_ranks_sharing_the_gpu():
    """The 2-rank code path on a 1-GPU box: both ranks bind GPU 0 and the group is gloo (MNRF_SHARE_GPU=1, a test aid:
    RCCL refuses two ranks on one device).  Everything but the transport is what the 8-GPU run executes: distinct ranks
    and batches, interleaved tiles, gather on rank 0, bucket all-reduce issued from the backward hooks, in-place table
    gradients, identical parameters after the step."""
    _run(2, MNRF_SHARE_GPU="1")


@pytest.mark.parametrize("mode,opt", [("skip", "torch"), ("sync", "torch"), ("skip", "flat")])
def test_guard_trip_on_one_of_two_ranks(mode, opt):
    """ADVICE r3 (medium), executed with two ranks: ONE rank's batch leaves the range of the split arithmetic.  "skip": the
    optimizer update is skipped on BOTH ranks (the flag is all-reduced on the device), the next step pins every model on every
    rank; "sync": both ranks recompute the step on the exact kernels (collective decision, the first pass's buckets discarded).
    Either way the ranks end with identical, finite weights and keep issuing the same collectives."""
    _run(2, worker=GUARD_WORKER, MNRF_SHARE_GPU="1", MNRF_GUARD_MODE=mode, MNRF_TEST_OPT=opt)


PINNED_WORKER = GUARD_WORKER.split("before = [q.detach()")[0] + textwrap.dedent('''
    # ADVICE r4 (medium): precision pinning OUTSIDE train_step is rank-local -- a validation pass on rank 0 that trips the guard
    # pins rank 0 only.  Rank 0 then has no split model left; in skip mode it used to skip the guard-flag all-reduce the other
    # ranks still issued (mismatched collectives).  Now every rank issues it whatever its state, and ranks in different states
    # converge: every model pinned on every rank from the next step on, identical weights, no hang.
    if rank == 0:
        MN.pin_fp32(system)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):
            training.train_step(system, opt, *batch(False))
        torch.cuda.synchronize()
    assert all(MN.precision_of(m) == "fp32" for m in system.models.values()), "ranks in different states must converge"
    c = checksum()
    assert all(abs(x - c[0]) <= 1e-9 * max(1.0, abs(c[0])) for x in c), c
    assert all(bool(torch.isfinite(q).all()) for q in system.parameters())
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank)
''')


@pytest.mark.parametrize("mode", ["skip", "sync"])
def test_one_rank_pinned_outside_the_training_step(mode):
    _run(2, worker=PINNED_WORKER, MNRF_SHARE_GPU="1", MNRF_GUARD_MODE=mode, MNRF_TEST_OPT="flat")


def test_bench_with_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2` end to end (launcher path, weak + strong scaling legs, training step with the all-reduce,
    the hash-grid leg across ranks) on one GPU shared by both ranks; the numbers are void, the line's shape is not."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MNRF_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = _share_env(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2, lines                      # rank 0 alone prints: the headline line at once, the complete one at the end
    first, line = json.loads(lines[0]), json.loads(lines[1])
    assert first["line"] == "headline" and line["line"] == "complete" and first["value"] == line["value"] and "train_step" not in first
    assert line["n_gpus"] == 2 and line["config"]["rccl_world_size"] == 2 and line["config"]["collective_backend"].startswith("gloo")
    assert line["scaling"] == "weak" and line["value"] > 0 and "cpu_baseline" not in line       # rank 0 at N = 1 only
    ss = line["strong_scaling"]
    assert ss["rays_per_s"] > 0 and 0 < ss["rays_of_rank0"] < 640000
    cm = ss["clustered_mask"]
    assert len(cm["reflected_rays_per_rank"]) == 2 and sum(cm["primary_rays_per_rank"]) == 640000.0
    assert line["train_step"]["ms_per_step"] > 0 and line["train_step"]["allreduce"]["buckets_in_place"] == 2
    hg = line["hash_grid_variant"]
    assert hg["n_gpus"] == 2 and hg["train_ms_per_step"] > 0 and hg["allreduce"] is not None


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` without a launcher re-executes under torch.distributed.run; here N = 1 with the
    collectives forced and the launcher path taken (MNRF_BENCH_SPAWN=1), so that the RCCL legs (strong_scaling, train_step all-reduce) run and are reported."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MNRF_FORCE_COLLECTIVES="1", MNRF_BENCH_SPAWN="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_world_size"] == 1 and line["config"]["collective_backend"].startswith("nccl")
    assert line["strong_scaling"]["rays_per_s"] > 0 and line["train_step"]["ms_per_step"] > 0
    cm = line["strong_scaling"]["clustered_mask"]          # load balance with a clustered (25 % rectangle) mirror mask
    assert cm["reflected_rays_per_rank"] == [160000.0] and cm["primary_rays_per_rank"] == [640000.0]
    assert cm["time_imbalance_max_over_mean"] == 1.0 and cm["frame_s_per_rank"][0] > 0
    assert line["train_step"]["allreduce_bytes_per_step"] > 0 and line["train_step"]["allreduce"]["buckets_in_place"] == 2
    assert line["train_step"]["config3_64_plus_192"]["samples_per_ray"] == 256      # BASELINE config 3 as worded
    hg = line["hash_grid_variant"]                                                    # config 5 with its collectives
    assert hg["allreduce"]["messages_per_step"] == 4 and hg["allreduce"]["bytes_per_step"] > 90e6 and hg["train_ms_per_step"] > 0
    assert 0 < hg["table_gradient_sparsity"]["touched_fraction"][1] < 1.0
    # round 6: the contract's training legs run on the static route across ranks; the step captured WITH its collectives is the
    # optional last leg, measured after the complete line went out (two "complete" lines: the last one carries it)
    assert line["train_step"]["route"] == "static" and line["line"] == "complete"
    g = line["train_step"]["graph_across_ranks"]
    assert g["route"] == "graph" and g["collectives_in_graph"] is True and g["ms_per_step"] > 0, g
    assert sum(l.startswith("{") and '"line": "complete"' in l for l in r.stdout.splitlines()) == 2
