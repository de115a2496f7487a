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


# ---- flat gradient buckets of the field modules (dist._flat_bucket / attach_overlap), world size 2 on gloo
class _FakeFieldFn(torch.autograd.Function):
    """Stands for autograd.FieldFn on the CPU: its backward hands autograd VIEWS of one flat buffer (32 parameters in
    state_dict order) and leaves that buffer on the module, like autograd._Pending.finish does."""

    @staticmethod
    def forward(ctx, module, fill, *params):
        ctx.module, ctx.fill, ctx.n = module, fill, len(params)
        return sum(q.sum() for q in params) * 0.0

    @staticmethod
    def backward(ctx, g):
        from mirror_nerf_amd.weights import PARAM_NAMES, PARAM_SHAPES
        sizes = [int(torch.Size(PARAM_SHAPES[n]).numel()) for n in PARAM_NAMES]
        flat = torch.full((sum(sizes),), ctx.fill)
        views, off = [], 0
        for n, k in zip(PARAM_NAMES, sizes):
            views.append(flat[off:off + k].view(PARAM_SHAPES[n]))
            off += k
        ctx.module.__dict__["_mnrf_flat_grad"] = flat
        return (None, None, *views)


def _bucket_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MNRF_FORCE_COLLECTIVES"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import mirror_nerf_amd as M
        from mirror_nerf_amd.weights import params_of
        torch.manual_seed(0)
        mods = [M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True) for _ in range(2)]
        extra = torch.nn.Parameter(torch.zeros(3))          # a parameter outside the field modules: generic path
        ov = D.attach_overlap(mods)
        assert len(ov) == 2
        res = []
        for step in range(2):
            for m in mods:
                for p_ in m.parameters():
                    p_.grad = None
            extra.grad = None
            loss = sum(_FakeFieldFn.apply(m, float(10 * i + rank + 1 + step), *params_of(m)) for i, m in enumerate(mods)) \
                + (extra * float(rank + 1)).sum()
            loss.backward()
            started = [o.work is not None for o in ov]      # the hooks issued the all-reduces from inside the backward pass
            flats = [m.__dict__["_mnrf_flat_grad"] for m in mods]
            D.allreduce_gradients([q_ for m in mods for q_ in params_of(m)] + [extra], modules=mods)
            alias = all(D._flat_bucket(m) is f for m, f in zip(mods, flats))
            vals = [float(m.sigma.weight.grad[0, 0]) for m in mods] + [float(mods[1].xyz_encoding_5[0].weight.grad[3, 7])]
            res.append((started, alias, vals, extra.grad.tolist()))
        # ranks that DIVERGE: on rank 1 the second module does not take part in the loss (its hooks never fire, its .grads
        # stay None), on rank 0 it does and its bucket goes out from inside the backward pass.  Every rank must still send
        # one message per module, of one size, in one order -- rank 1 sends zeros for the module it has no gradient of
        for m in mods:
            for p_ in m.parameters():
                p_.grad = None
        take = mods if rank == 0 else mods[:1]
        loss = sum(_FakeFieldFn.apply(m, float(100 * (i + 1) + rank), *params_of(m)) for i, m in enumerate(take))
        loss.backward()
        D.allreduce_gradients([q_ for m in mods for q_ in params_of(m)], modules=mods)
        res.append([float(m.sigma.weight.grad[0, 0]) for m in mods])
        # a module whose .grads do not alias its flat buffer (autograd copied): generic path, still correct
        m = mods[0]
        for p_ in m.parameters():
            p_.grad = torch.full_like(p_, float(rank))
        D.allreduce_gradients(params_of(m), modules=[m])
        res.append(float(m.sigma.bias.grad[0]))

        # ---- a RECOMPUTED step (training.train_step after a range-guard trip): the first pass's buckets went out
        # from inside backward; reset_overlap() drains and discards them, the second pass is reduced normally
        def backward_pass(fill):
            for m_ in mods:
                for p_ in m_.parameters():
                    p_.grad = None
            sum(_FakeFieldFn.apply(m_, float(fill + 10 * i + rank), *params_of(m_)) for i, m_ in enumerate(mods)).backward()
        backward_pass(1000.0)
        assert all(o.work is not None for o in ov)
        raised = False
        try:                       # a second backward pass without reset: refused, not silently reduced twice
            sum(_FakeFieldFn.apply(m_, 0.0, *params_of(m_)) for m_ in mods).backward()
        except RuntimeError as e:
            raised = "second backward pass" in str(e)
        D.reset_overlap()
        assert all(o.work is None and o.left == len(o.params) for o in ov)
        backward_pass(1.0)
        D.allreduce_gradients([q_ for m_ in mods for q_ in params_of(m_)], modules=mods)
        res.append((raised, [float(m_.sigma.weight.grad[0, 0]) for m_ in mods]))
        # ---- gradient accumulation under no_overlap(): two passes, nothing sent from the hooks, one reduction at the end
        for m_ in mods:
            for p_ in m_.parameters():
                p_.grad = None
        with D.no_overlap():
            for fill in (1.0, 2.0):
                sum(_FakeFieldFn.apply(m_, float(fill + rank), *params_of(m_)) for m_ in mods).backward()
                assert all(o.work is None for o in ov)
        D.allreduce_gradients([q_ for m_ in mods for q_ in params_of(m_)], modules=mods)
        res.append([float(m_.sigma.weight.grad[0, 0]) for m_ in mods])
        # ---- diverging ranks AND a collective of the caller's own between backward() and allreduce_gradients() (train_step's
        # range-guard flag): on rank 0 both buckets left from inside the backward pass, on rank 1 none did (the module at the
        # head of the fixed order never became ready).  issue_pending() sends the rest first, so the flag is the THIRD collective
        # on both ranks; without it rank 1 would put it first and the sequences would cross
        for m_ in mods:
            for p_ in m_.parameters():
                p_.grad = None
        take = mods if rank == 0 else mods[:1]
        sum(_FakeFieldFn.apply(m_, float(300 * (i + 1) + rank), *params_of(m_)) for i, m_ in enumerate(take)).backward()
        sent_in_backward = [o.work is not None for o in ov]
        D.issue_pending()
        sent_before_flag = [o.work is not None for o in ov]
        flag = torch.tensor([float(rank)])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        D.allreduce_gradients([q_ for m_ in mods for q_ in params_of(m_)], modules=mods)
        res.append((sent_in_backward, sent_before_flag, float(flag), [float(m_.sigma.weight.grad[0, 0]) for m_ in mods]))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_buckets_and_overlapped_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        res = out[rank]
        for step in range(2):
            started, alias, vals, eg = res[step]
            assert started == [True, True]           # issued during backward (post-accumulate-grad hooks)
            assert alias                             # reduced in place: every .grad is still a view of the flat buffer
            # mean over ranks of (10 i + rank + 1 + step)
            assert vals == [1.5 + step, 11.5 + step, 11.5 + step], vals
            assert eg == [1.5] * 3
        assert res[2] == [100.5, 100.0], res[2]        # mean(100, 101); mean(200 on rank 0, nothing on rank 1)
        assert res[3] == 0.5
        assert res[4] == (True, [1.5, 11.5]), res[4]      # the recomputed pass only: mean(1 + 10 i + rank)
        assert res[5] == [4.0, 4.0], res[5]               # (1 + 2) + mean over ranks of 2 * rank
        assert res[6] == ([rank == 0] * 2, [True, True], 1.0, [300.5, 300.0]), res[6]


# ---- hash-grid model (BASELINE config 5 "... 8xMI355X"): table gradient reduced in place, MLP gradients as one blob
def _tcnn_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MNRF_FORCE_COLLECTIVES"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import mirror_nerf_amd as M
        from mirror_nerf_amd.weights import params_of
        torch.manual_seed(0)
        m = M.MirrorNeRFTcnn(encoding="hashgrid", bound=1.0, predict_normal=True, predict_mirror_mask=True)
        table = m.encoder.embeddings
        res = []
        # (a) plain path: .grads set by hand; rank 1 has no gradient for one MLP tensor and none for the table
        rows = torch.tensor([3, 70000, 5000000]) + rank
        if rank == 0:
            table.grad = torch.zeros_like(table)
            table.grad[rows] = 2.0
        for i, p_ in enumerate(m.mlp_params()):
            p_.grad = None if (rank == 1 and i == 4) else torch.full_like(p_, float(rank + 1 + i))
        ptr = table.grad.data_ptr() if table.grad is not None else None
        D.allreduce_gradients(params_of(m), modules=[m])
        res.append((table.grad.data_ptr() == ptr if ptr is not None else True, float(table.grad[3, 0]), float(table.grad[4, 1]),
                    int((table.grad != 0).sum()), [float(p_.grad.reshape(-1)[0]) for p_ in m.mlp_params()]))
        # (b) overlapped: the all-reduces go out from the post-accumulate hooks of a real backward pass
        ov = D.attach_overlap([m])
        assert len(ov) == 1
        for p_ in m.parameters():
            p_.grad = None
        loss = table[rows].sum() * float(rank + 1) + sum(p_.sum() for p_ in m.mlp_params()) * float(10 * (rank + 1))
        loss.backward()
        started = ov[0].work is not None and len(ov[0].work) == 2
        tg = table.grad
        D.allreduce_gradients(params_of(m), modules=[m])
        res.append((started, table.grad is tg, float(table.grad[3, 0]), float(table.grad[4, 0]), float(m.sigma_net[0].weight.grad[0, 0]),
                    float(m.is_mirror_net[2].bias.grad[0])))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_hash_grid_model_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcnn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank in (0, 1):
        a, b = out[rank]
        in_place, g3, g4, nnz, mlp = a
        assert in_place and g3 == 1.0 and g4 == 0.0 and nnz == 6      # mean(2, 0) on rank 0's three rows x 2 features
        want = [1.5 + i for i in range(11)]
        want[4] = 2.5                                                  # (5 + 0) / 2: rank 1 sent zeros for the tensor it has no gradient of
        assert mlp == want, mlp
        started, same_tensor, t3, t4, w, bias = b
        assert started and same_tensor
        assert (t3, t4) == (0.5, 1.0)       # row 3: rank 0 only (factor 1, / 2); row 4: rank 1 only (factor 2, / 2)
        assert w == 15.0 and bias == 15.0


# ---- training.train_step's collective sequence with FOUR ranks, a guard trip on rank 2 and a rank pinned outside the step
class _FakeEvent:
    def query(self):
        return True

    def synchronize(self):
        pass

    def record(self):
        pass


def _train_step_worker(rank, ws, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MNRF_FORCE_COLLECTIVES"] = "0"
    os.environ["MNRF_GUARD_MODE"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import mirror_nerf_amd as M
        from mirror_nerf_amd import mirror_nerf as MN, training as T
        from mirror_nerf_amd.weights import params_of
        assert T.GUARD_MODE == mode
        torch.manual_seed(0)

        class System(torch.nn.Module):          # stands for NeRFSystem on the CPU: two field modules evaluated through _FakeFieldFn
            def __init__(self):
                super().__init__()
                self.hparams = T.default_hparams()
                self.nerf_coarse = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
                self.nerf_fine = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
                self.models = {"coarse": self.nerf_coarse, "fine": self.nerf_fine}

            def forward(self, rays, extra):
                return {"loss": sum(_FakeFieldFn.apply(m, float(1 + rank), *params_of(m)) for m in self.models.values()) + rays.sum() * 0.0}

        system = System()
        words = {"now": [0, 0]}                 # the guard words the fake device would hold after this step's launches

        def begin(sysm):                        # mirror_nerf.guard_async_begin without a device
            mods = [m for m in sysm.models.values() if MN.precision_of(m).startswith("split")]
            if not MN.GUARD or not mods:
                return None
            w = torch.tensor([words["now"][i] for i, m in enumerate(sysm.models.values()) if m in mods], dtype=torch.int32)
            return mods, w.clone(), _FakeEvent(), w

        def check(sysm):                        # mirror_nerf.check_guard (sync mode)
            tripped = False
            for i, m in enumerate(sysm.models.values()):
                if MN.precision_of(m).startswith("split") and words["now"][i]:
                    m.__dict__["_mnrf_precision"] = "fp32"
                    tripped = True
            return tripped
        MN.guard_async_begin, MN.check_guard = begin, check

        class Opt:                              # takes found_inf like torch's fused Adam / FlatAdam
            mnrf_found_inf = True
            grad_scale = found_inf = None
            log = []

            def zero_grad(self, set_to_none=True):
                for p_ in system.parameters():
                    p_.grad = None

            def step(self):
                self.log.append(None if self.found_inf is None else float(self.found_inf))
        opt = Opt()
        D.attach_overlap(system.models.values())
        seq = []
        real = dist.all_reduce

        def logged(t, *a, **k):                 # every all-reduce this rank issues: (elements, async?)
            seq.append((int(t.numel()), bool(k.get("async_op", False))))
            return real(t, *a, **k)
        dist.all_reduce = torch.distributed.all_reduce = logged
        rays, target, gt = torch.zeros(8, 8), torch.zeros(8, 3), torch.zeros(8)
        loss_fn = lambda res, t, g: res["loss"]  # noqa: E731
        states = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for step in range(5):
                words["now"] = [0, 1] if (step == 1 and rank == 2) else [0, 0]      # rank 2's fine model leaves the range in step 1
                T.train_step(system, opt, rays, target, gt, loss_fn)
                states.append([MN.precision_of(m) for m in system.models.values()])
        q.put((rank, seq, states, list(opt.log)))
    finally:
        dist.destroy_process_group()


def _pinned_rank_worker(rank, ws, port, q, mode):
    """rank 0 enters training with its models already pinned (a validation pass tripped its guard)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MNRF_FORCE_COLLECTIVES"] = "0"
    os.environ["MNRF_GUARD_MODE"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import mirror_nerf_amd as M
        from mirror_nerf_amd import mirror_nerf as MN, training as T
        from mirror_nerf_amd.weights import params_of
        torch.manual_seed(0)

        class System(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.hparams = T.default_hparams()
                self.nerf_coarse = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
                self.models = {"coarse": self.nerf_coarse}

            def forward(self, rays, extra):
                return {"loss": _FakeFieldFn.apply(self.nerf_coarse, float(1 + rank), *params_of(self.nerf_coarse)) + rays.sum() * 0.0}
        system = System()

        def begin(sysm):
            mods = [m for m in sysm.models.values() if MN.precision_of(m).startswith("split")]
            if not mods:
                return None
            w = torch.zeros(len(mods), dtype=torch.int32)
            return mods, w.clone(), _FakeEvent(), w
        MN.guard_async_begin, MN.check_guard = begin, (lambda sysm: False)

        class Opt:
            mnrf_found_inf = True
            grad_scale = found_inf = None
            log = []

            def zero_grad(self, set_to_none=True):
                for p_ in system.parameters():
                    p_.grad = None

            def step(self):
                self.log.append(None if self.found_inf is None else float(self.found_inf))
        opt = Opt()
        D.attach_overlap(system.models.values())
        if rank == 0:
            MN.pin_fp32(system)                 # outside train_step, on this rank only
        seq = []
        real = dist.all_reduce

        def logged(t, *a, **k):
            seq.append((int(t.numel()), bool(k.get("async_op", False))))
            return real(t, *a, **k)
        dist.all_reduce = torch.distributed.all_reduce = logged
        rays, target, gt = torch.zeros(8, 8), torch.zeros(8, 3), torch.zeros(8)
        states = []
        for _ in range(3):
            T.train_step(system, opt, rays, target, gt, lambda res, t, g: res["loss"])
            states.append(MN.precision_of(system.nerf_coarse))
        q.put((rank, seq, states, list(opt.log)))
    finally:
        dist.destroy_process_group()


def _spawn(target, ws, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, ws, port, q) + args) for r in range(ws)]
    for p in procs:
        p.start()
    out = dict((r[0], r[1:]) for r in (q.get(timeout=240) for _ in procs))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def test_train_step_collective_sequence_with_four_ranks_and_a_trip_on_rank_two():
    """training.train_step with four ranks on gloo, kernels replaced by stand-ins (a fake field function whose
    backward hands out views of one flat buffer, fake guard words).  Rank 2's fine model leaves the range in step 1.  Every rank
    issues the IDENTICAL sequence of all-reduces in every step (two async buckets, then the three guard flags, blocking); the
    tripping step's update is skipped on every rank (found_inf = 1 everywhere); from step 2 on every model of every rank is pinned."""
    for mode in ("skip", "sync"):
        out = _spawn(_train_step_worker, 4, mode)
        seqs = [out[r][0] for r in range(4)]
        assert all(s == seqs[0] for s in seqs), (mode, seqs)
        n_flat = seqs[0][0][0]                                  # one flat bucket per model (662 152 floats), async, then the flags
        assert n_flat > 600000 and seqs[0][:3] == [(n_flat, True), (n_flat, True), (3, False)], seqs[0][:6]
        for r in range(4):
            _seq, states, log = out[r]
            assert states[0] == ["split", "split"]
            assert states[2] == ["fp32", "fp32"] and states[4] == ["fp32", "fp32"], (mode, r, states)
            if mode == "skip":
                assert log[0] == 0.0 and log[1] == 1.0, (r, log)      # step 1's update vetoed on EVERY rank
                assert log[3] in (None, 0.0) and log[4] in (None, 0.0), (r, log)


def test_a_rank_pinned_outside_the_step_keeps_the_collectives_matched():
    """rank 0 has no split model left when training starts.  It used to skip the guard-flag all-reduce
    that the other ranks issue (the next collective on rank 0 was a bucket of another size: a hang under RCCL)."""
    for mode in ("skip", "sync"):
        out = _spawn(_pinned_rank_worker, 2, mode)
        assert out[0][0] == out[1][0], (mode, out[0][0], out[1][0])
        assert all(n == 3 and not a for (n, a) in out[0][0][1::2]), out[0][0]      # every step: one bucket, then the three flags
        for r in (0, 1):
            assert out[r][1][-1] == "fp32", (mode, r, out[r][1])                   # the ranks converged
            assert all(v in (None, 0.0) for v in out[r][2]), out[r][2]              # no update was vetoed: nothing tripped
