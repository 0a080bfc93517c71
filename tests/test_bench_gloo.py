"""bench.py's world > 1 branches end to end on the host: two gloo ranks walk `bench.main()` (warm-up, timed steps of a whole
8-iteration cycle, the roofline iteration rank 0 profiles and its lock-step twin on the other ranks, the launch-table iteration,
flush + barrier + teardown) with the tests' kernel emulation installed, and every rank must issue the SAME sequence of collectives
-- the property an RCCL run needs in order not to hang (reference: loader.py:13-19, worker.py:88-96; `backend="nccl"` itself needs
two devices)."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir, check):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "LCGAN_DIST_BACKEND": "gloo", "LCGAN_BENCH_DEVICE": "cpu",
                       "LCGAN_DDP_CHECK": "1" if check else "0"})
    torch.set_num_threads(2)
    from oracle.hip_emulation import EmulatedKernels
    from tests.helpers import install_backend
    from lcgan_amd.optim import DataParallel

    class Emu(EmulatedKernels):
        def prof_dump(self, path):
            open(path, "w").close()
    install_backend(Emu())
    DataParallel.BUCKET_BYTES = 1 << 20
    log = []
    real = {n: getattr(dist, n) for n in ("all_reduce", "broadcast", "barrier")}

    def wrap(name):
        def f(*a, **kw):
            t = a[0] if a and torch.is_tensor(a[0]) else None
            log.append((name, None if t is None else t.numel()))
            return real[name](*a, **kw)
        return f
    for n in real:
        setattr(dist, n, wrap(n))
    import bench
    table = os.path.join(out_dir, f"table{rank}.csv")
    sys.argv = ["bench.py", "--gpus", str(world), "--res", "16", "--batch", "4", "--steps", "8", "--warmup", "1", "--epoch-type", "cycle",
                "--dtype", "f32", "--launch-table", table]
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"log": log, "stdout": buf.getvalue()}, f)


def _run(tmp_path, check):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), check), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    assert r[0]["log"] == r[1]["log"], "the ranks issued different collective sequences"
    names = [n for n, _ in r[0]["log"]]
    # 11 iterations (1 warm-up + 8 timed + roofline + launch table), each reducing G and D gradients in buckets
    assert names.count("all_reduce") >= 2 * 11 and names.count("barrier") >= 3 and names.count("broadcast") > 0
    line = json.loads(r[0]["stdout"].strip().splitlines()[-1])           # rank 0 prints the ONE JSON line, the others nothing
    assert r[1]["stdout"].strip() == ""
    assert line["n_gpus"] == 2 and line["steps"] == 8 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert os.path.exists(tmp_path / "table0.csv") and not os.path.exists(tmp_path / "table1.csv")
    return r[0]["log"]


def test_bench_two_ranks_gloo(tmp_path):
    _run(tmp_path, check=False)


def test_bench_two_ranks_gloo_check_mode(tmp_path):
    """the same walk with DataParallel's debug verification on (LCGAN_DDP_CHECK=1): one extra int32 all-reduce per sync, no mismatch"""
    log = _run(tmp_path, check=True)
    assert any(n == "all_reduce" and sz is not None and sz < 4096 for n, sz in log)
