"""N > 1 path on the MI355X: two ranks sharing the one GPU of the test box (gloo carries the collective: RCCL refuses two
ranks on one device), each running the real HIP kernels and lcgan_amd.optim.DataParallel.sync_gradients.  Checks that the
gradient bucket ends up identical on both ranks, equals the mean of single-rank runs on the same half-batches, and that the
replicas stay in lock-step after the multi-tensor Adam."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _feed(w, Bl, res, dev, rank):
    from oracle.weights import seeded_tensor
    from tests.helpers import FixedFeed
    feed = FixedFeed(w, Bl, res, dev)
    feed.z = [seeded_tensor((Bl, 64), 500 + 10 * rank + i).to(dev) for i in range(4)]
    feed.real = tuple(seeded_tensor((Bl, 3, res, res), 600 + 10 * rank + i, "uniform_pm1").to(dev) for i in range(3))
    return feed


def _rank_main(rank, world, port, out_dir, backend="gloo"):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from lcgan_amd import config
    from tests.helpers import seeded_worker
    dev_index = rank if backend == "nccl" else 0                  # RCCL: one device per rank; gloo: both ranks on the one GPU
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    config.set_feature_dtype(torch.float32)
    res, Bl, dev = 32, 4, f"cuda:{dev_index}"
    w = seeded_worker(res, Bl, dev, gpus=world)
    _feed(w, Bl, res, dev, rank)
    w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
    captured = {}
    real_step = w.g_optimizer.step

    def step():
        captured.update({k: p.grad.detach().cpu().clone() for k, p in w.generator.module.named_parameters()})
        real_step()
    w.g_optimizer.step = step
    loss_v = float(w.train_generator(1))
    w.flush()
    torch.save({"grads": captured, "params": {k: v.cpu() for k, v in w.generator.module.state_dict().items()}, "loss": loss_v},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path):
    from lcgan_amd import config
    from tests.helpers import seeded_worker
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    single = []
    with config.feature_dtype_as(torch.float32):
        for r in range(world):
            w = seeded_worker(32, 4, "cuda:0")
            _feed(w, 4, 32, "cuda:0", r)
            w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
            w.g_optimizer.step = lambda: None
            w.train_generator(1)
            single.append({k: p.grad.detach().cpu().clone() for k, p in w.generator.module.named_parameters()})
    worst = 0.0
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k                    # identical bucket on both ranks
        mean = (single[0][k] + single[1][k]) / 2
        err = float((r0["grads"][k] - mean).norm() / mean.norm().clamp_min(1e-30))
        worst = max(worst, err)
    assert worst <= 5e-3, worst                                                  # atomics order / kink noise only
    for k in r0["params"]:
        if k.startswith("avg_latent"):
            assert not torch.equal(r0["params"][k], r1["params"][k])             # per-rank statistics (broadcast_buffers=False, worker.py:90)
        else:
            assert torch.equal(r0["params"][k], r1["params"][k]), k              # replicas in lock-step after Adam
    assert r0["loss"] != r1["loss"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two devices (the test box has one)")
def test_two_ranks_rccl(tmp_path):
    """The same two-rank G step with backend "nccl" (= RCCL over xGMI), one device per rank: exercised the first time the test
    box has two GPUs.  Both ranks must hold the identical reduced bucket and identical parameters after Adam."""
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), "nccl"), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    for k in r0["params"]:
        if not k.startswith("avg_latent"):
            assert torch.equal(r0["params"][k], r1["params"][k]), k
