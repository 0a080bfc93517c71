"""N > 1 path on CPU: two gloo ranks, each with its own half of the global batch, must end up with the mean gradient in
every .grad, identical parameters after Adam, and leave unused parameters (grad None) untouched -- the semantics of the
reference's DistributedDataParallel(find_unused_parameters=True) + torch.optim.Adam (worker.py:88-110)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    import lcgan_amd.kernels as KM
    from lcgan_amd import config
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend, FixedFeed, seeded_worker
    from oracle.weights import seeded_tensor
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    install_backend(EmulatedKernels())
    config.set_feature_dtype(torch.float32)
    res, Bl = 16, 2
    from lcgan_amd.optim import DataParallel
    DataParallel.BUCKET_BYTES = 1 << 20                   # this 16 x 16 network is small: 1 MB buckets give the shared layers several
    w = seeded_worker(res, Bl, "cpu", gpus=world)
    assert w.local_batch_size == Bl
    feed = FixedFeed(w, Bl, res, "cpu")
    feed.z = [seeded_tensor((Bl, 64), 500 + 10 * rank + i) for i in range(4)]           # different latents per rank
    feed.real = tuple(seeded_tensor((Bl, 3, res, res), 600 + 10 * rank + i, "uniform_pm1") for i in range(3))
    w.requires_grad(w.generator, False), w.requires_grad(w.discriminator, True)
    captured = {}
    real_step = w.d_optimizer.step

    def step():
        captured.update({k: (None if p.grad is None else p.grad.clone()) for k, p in w.discriminator.module.named_parameters()})
        real_step()
    w.d_optimizer.step = step
    # the reduction is bucketed and starts INSIDE the backward pass (post-accumulate hooks): record where each bucket was launched
    D = w.discriminator
    launches, in_sync = [], [False]
    real_launch, real_sync = D._launch, D.sync_gradients

    def launch(b):
        launches.append((D._buckets.index(b), "sync" if in_sync[0] else "hook", any(p.grad is not None for p in b.params)))
        real_launch(b)

    def sync(async_op=False):
        in_sync[0] = True
        try:
            return real_sync(async_op=async_op)
        finally:
            in_sync[0] = False
    D._launch, D.sync_gradients = launch, sync
    loss_v = w.train_discriminator(1)                     # odd + R1: projection heads unused -> grad None
    w.flush()                                             # the all-reduce wait + Adam are postponed when N > 1
    assert len(D._buckets) >= 4, len(D._buckets)
    assert sorted(i for i, _, _ in launches) == list(range(len(D._buckets)))          # every bucket exactly once
    hooked = [i for i, where, used in launches if where == "hook" and used]
    assert len(hooked) >= 3, launches                     # reductions were issued while the backward pass was still running
    assert hooked == sorted(hooked), launches             # ... in bucket (= reverse registration) order
    assert any(not used for _, _, used in launches), launches     # the projection heads' buckets carried nothing and were skipped
    torch.save({"grads": captured, "params": {k: v.clone() for k, v in w.discriminator.module.state_dict().items()}, "loss": float(loss_v)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _single(rank_seed_offsets, out):
    """reference result: the same two half-batches processed by ONE process, gradients averaged by hand."""
    import lcgan_amd.kernels as KM
    from lcgan_amd import config
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend, FixedFeed, seeded_worker
    from oracle.weights import seeded_tensor
    install_backend(EmulatedKernels())
    config.set_feature_dtype(torch.float32)
    res, Bl = 16, 2
    grads = []
    for r in rank_seed_offsets:
        w = seeded_worker(res, Bl, "cpu")
        feed = FixedFeed(w, Bl, res, "cpu")
        feed.z = [seeded_tensor((Bl, 64), 500 + 10 * r + i) for i in range(4)]
        feed.real = tuple(seeded_tensor((Bl, 3, res, res), 600 + 10 * r + i, "uniform_pm1") for i in range(3))
        w.requires_grad(w.generator, False), w.requires_grad(w.discriminator, True)
        w.d_optimizer.step = lambda: None
        w.train_discriminator(1)
        grads.append({k: (None if p.grad is None else p.grad.clone()) for k, p in w.discriminator.module.named_parameters()})
    install_backend(None)
    return grads


def test_two_rank_gradient_mean_and_adam(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    single = _single([0, 1], tmp_path)
    for k in r0["grads"]:
        g0, g1 = r0["grads"][k], r1["grads"][k]
        if single[0][k] is None:
            assert g0 is None and g1 is None, k                     # unused parameter stays None on every rank
            continue
        mean = (single[0][k] + single[1][k]) / 2
        assert torch.allclose(g0, mean, rtol=1e-4, atol=1e-6 * float(mean.abs().max())), k
        assert torch.equal(g0, g1), k                               # all-reduce leaves identical gradients
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k     # replicas stay in lock-step after Adam
    assert r0["loss"] != r1["loss"]                                 # each rank saw its own half of the batch


def _iter_main(rank, world, port, out_dir, defer):
    import lcgan_amd.kernels as KM
    from lcgan_amd import config, loader
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend, FixedFeed, make_args, seeded_worker
    from oracle.weights import seeded_tensor
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    install_backend(EmulatedKernels())
    config.set_feature_dtype(torch.float32)
    res, Bl = 16, 2
    torch.manual_seed(0)                                      # the EMA copy starts from the (random) constructor weights
    w = seeded_worker(res, Bl, "cpu", gpus=world)
    w.ema = type(w.ema)(w.generator, w.generator_ema, 0.9, 0)     # re-copy the seeded weights, visible decay
    feed = FixedFeed(w, Bl, res, "cpu")
    feed.z = [seeded_tensor((Bl, 64), 700 + 10 * rank + i) for i in range(4)]
    feed.real = tuple(seeded_tensor((Bl, 3, res, res), 800 + 10 * rank + i, "uniform_pm1") for i in range(3))
    if not defer:                                             # reference behaviour: wait + step right after the backward
        def sync_after(key, model, optimizer):
            model.sync_gradients(async_op=False)
            optimizer.step()
        w._after_backward = sync_after
    args = make_args(res, Bl * world)
    for epoch in (0, 1, 2):
        loader.train_iteration(w, args, epoch)
    w.flush()
    state = {"g": {k: v.clone() for k, v in w.generator.module.state_dict().items()},
             "d": {k: v.clone() for k, v in w.discriminator.module.state_dict().items()},
             "ema": {k: v.clone() for k, v in w.generator_ema.module.state_dict().items()}}
    torch.save(state, os.path.join(out_dir, f"it_rank{rank}_{int(defer)}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_deferred_allreduce_equals_synchronous(tmp_path):
    """Three iterations on two ranks: postponing the all-reduce wait + Adam + EMA behind the other network's forward
    (worker.WORKER._after_backward) must give bit-identical parameters to waiting immediately, and identical replicas."""
    world = 2
    for defer in (True, False):
        mp.spawn(_iter_main, args=(world, _free_port(), str(tmp_path), defer), nprocs=world, join=True)
    a0, a1 = (torch.load(tmp_path / f"it_rank{r}_1.pt") for r in range(world))
    b0 = torch.load(tmp_path / "it_rank0_0.pt")
    for net in ("g", "d", "ema"):
        for k in a0[net]:
            if k.startswith("avg_latent"):
                continue                                        # per-rank buffers
            assert torch.equal(a0[net][k], a1[net][k]), (net, k)
            assert torch.equal(a0[net][k], b0[net][k]), (net, k)


def _guard_main(rank, world, port, out_dir):
    """DataParallel's contract on a toy module: one backward per sync (a second one raises instead of corrupting a bucket in flight),
    zero_grad() clears an aborted step, check mode verifies the ranks' gradient sets before any collective and equals the unchecked path."""
    import lcgan_amd.kernels as KM
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend
    from lcgan_amd.optim import Adam, DataParallel
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    install_backend(EmulatedKernels())
    DataParallel.BUCKET_BYTES = 64 * 4                     # 64 floats per bucket: the 4 layers below land in 4 buckets

    def make(check):
        torch.manual_seed(0)
        net = torch.nn.Sequential(*[torch.nn.Linear(8, 8, bias=False) for _ in range(4)])
        dp = DataParallel(net, None, check=check)
        return net, dp, Adam(list(net.parameters()), lr=1e-2, on_zero_grad=dp.reset_reduction)

    x = torch.full((2, 8), float(rank + 1))
    # (1) one backward, one sync: mean gradients, identical with and without check mode
    grads = {}
    for check in (False, True):
        net, dp, opt = make(check)
        opt.zero_grad()
        dp(x).sum().backward()
        assert (len(dp._works) > 0) == (not check)         # check mode launches nothing from the hooks
        dp.sync_gradients()
        grads[check] = [p.grad.clone() for p in net.parameters()]
    for a, b in zip(grads[False], grads[True]):
        assert torch.equal(a, b)
    # (2) a second backward before the sync raises from the hook (both modes)
    for check in (False, True):
        net, dp, opt = make(check)
        opt.zero_grad()
        dp(x).sum().backward()
        with pytest.raises(RuntimeError, match="second backward"):
            dp(x).sum().backward()
        # (3) ... and the next step's zero_grad() recovers: buckets re-armed, reductions of the dead step drained
        opt.zero_grad()
        assert not dp._works and not dp._armed and all(not b.launched and b.fired == 0 for b in dp._buckets)
        dp(x).sum().backward()
        dp.sync_gradients()
        for a, p in zip(grads[False], net.parameters()):
            assert torch.allclose(a, p.grad)
    # (4) a rank-local grad=None set (rank 1 freezes a layer the others train): check mode raises on EVERY rank, naming the parameter
    net, dp, opt = make(True)
    if rank == 1:
        net[1].weight.requires_grad = False
    opt.zero_grad()
    dp(x).sum().backward()
    with pytest.raises(RuntimeError, match=r"ranks disagree.*1\.weight"):
        dp.sync_gradients()
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"guard{rank}.ok"), "w").close()


def test_data_parallel_guards(tmp_path):
    world = 2
    mp.spawn(_guard_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"guard{r}.ok").exists() for r in range(world))
