"""bench.py -- images/sec of the LC-GAN G+D training iteration on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full iteration in the reference's order (loader.py:45-54: train_generator -> ema_update -> train_discriminator)
of the odd+R1 type (`epoch % 8 == 1`), which is what BASELINE.json configs[1] names: 256x256, GLOBAL batch 32 (split over the
ranks like worker.py:35), bf16 feature maps / fp32 accumulation, synthetic data resident in HBM, reference-style random init.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# algorithmic dense-contraction work (2*MAC of conv + linear only) per image, SURVEY.md section 8 / BASELINE.md section 3
G_FWD = {256: 112.569e9, 512: 142.987e9, 1024: 173.756e9}
D_FWD = {256: 93.063e9, 512: 123.178e9, 1024: 153.344e9}
PEAK_BF16_DENSE = 2.5e15          # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_HBM = 8.0e12                 # ibid.: 8 TB/s HBM3E (6.3 TB/s achievable)


def _analytic_fwd_flops(res: int):
    """2*MAC of the conv layers for resolutions outside the SURVEY table (dry runs); within 1 % of the table at 256."""
    import math
    nb = int(math.log2(res)) - 2
    base = 32 if res == 1024 else 64 if res == 512 else 128
    g, cin, h = 0.0, 512, 4
    for i in range(nb):
        cout = min(base * 2 ** (nb - i - 1), 512)
        g += 2 * 9 * cin * cout * h * h + 2 * 9 * cout * cout * (2 * h) ** 2 + 2 * cin * cout * h * h + 2 * 9 * cin * 2 * h * h
        cin, h = cout, 2 * h
    g += 2 * 9 * cin * cin * h * h + 2 * cin * 3 * h * h
    d, h = 2 * 3 * base * res * res, res
    for i in range(nb):
        ci, co = min(base * 2 ** i, 512), min(base * 2 ** (i + 1), 512)
        d += 2 * 9 * ci * ci * h * h + 2 * 9 * ci * co * (h // 2) ** 2 + 2 * ci * co * (h // 2) ** 2
        h //= 2
    d += 2 * 9 * 513 * 512 * 16 + 2 * 8192 * 512
    return g, d


def flops_per_image(res: int, epoch: int) -> float:
    g, d = (G_FWD[res], D_FWD[res]) if res in G_FWD else _analytic_fwd_flops(res)
    if epoch % 2 == 0:
        return 10 * g + 18 * d + 15 * 0.072e9
    return 4 * g + (11 if epoch % 8 == 1 else 8) * d


def cpu_baseline(res: int, batch: int = 4, full_cycle: bool = False):
    """The CPU oracle (oracle/lcgan_ref.py, a port of the reference's fp32 PyTorch path) timed on the host cores at BASELINE
    configs[0]'s shape (256 x 256, batch 4; SURVEY 8(d) / BASELINE.md section 4).  Default: the bounded sample -- ONE warmed iteration of
    each type (even / odd+R1 / odd: G step + D step, fp32, no optimiser -- Adam + EMA are < 1 % of the CPU step), ~35 s of CPU work,
    from which the cycle mean (4 even + 1 R1 + 3 odd) follows.  full_cycle (`--cpu-baseline-cycle`, ~4 minutes): one warm-up cycle of
    8 iterations, then one timed cycle, as BASELINE.md section 4 words it.  `value` is the R1 iteration, the one the GPU line reports."""
    from oracle import lcgan_ref as O
    from oracle.weights import seeded_state, seeded_tensor
    torch.set_num_threads(min(os.cpu_count() or 1, 16))        # the CPU share of a one-GPU box
    GP, DP = seeded_state(O.g_param_shapes(res), 1001), seeded_state(O.d_param_shapes(res), 1002)

    def one(epoch, b):
        z = tuple(seeded_tensor((b, 64), 10 + i) for i in range(4))
        real = tuple(seeded_tensor((b, 3, res, res), 20 + i, "uniform_pm1") for i in range(3))
        t0 = time.perf_counter()
        O.g_step(GP, DP, res, epoch, z)
        O.d_step(GP, DP, res, epoch, z[:2], real)
        return time.perf_counter() - t0
    if full_cycle:
        for e in range(8):
            one(e, batch)                                      # warm-up cycle
        per = [one(8 + e, batch) for e in range(8)]            # epoch 8..15: 4 even, 1 odd+R1 (9), 3 odd
        t = {"even": sum(per[0::2]) / 4, "r1": per[1], "odd": (per[3] + per[5] + per[7]) / 3}
        cycle = sum(per) / 8
        sample = f"one warm-up cycle, then one timed 8-iteration cycle (epoch 8..15) at {res}x{res}, batch {batch}"
    else:
        one(1, 1)                                              # warm-up: thread pool, oneDNN primitives, allocator
        t = {"even": one(0, batch), "r1": one(1, batch), "odd": one(3, batch)}
        cycle = (4 * t["even"] + t["r1"] + 3 * t["odd"]) / 8
        sample = f"one warmed iteration per type at {res}x{res}, batch {batch}"
    return {"value": batch / t["r1"], "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample} (G step + D step, fp32, no optimiser): even {t['even']:.1f} s, odd+R1 {t['r1']:.1f} s, odd {t['odd']:.1f} s",
            "cycle_mean_value": batch / cycle, "seconds_per_type": {k: round(v, 2) for k, v in t.items()}}


def committed_traffic():
    """Memory-side bytes per launch of the dominant kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH_SIZE x 2 on gfx950: scripts/pmc_traffic.py -> profiles/r04_pmc_traffic.json).  PMC counters cannot be read
    inside a timed run, so the bench line quotes the profiled figure -- but only while the kernel sources of the running build hash to
    what the profiled build hashed to (`_srchash`); otherwise `traffic` is null and the file is named under `traffic_source` alone."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        fam = d["families"]["conv_halo"]
        from lcgan_amd.build import source_hash
        same = d.get("_srchash") is not None and d.get("_srchash") == source_hash()
        amp = (d.get("amplification") or {}).get("conv fwd/dgrad")
        return {"bytes_per_launch": fam["bytes_per_launch"] if same else None, "kernel": "conv_halo_kernel", "launches": fam["launches"],
                "source": "profiles/r04_pmc_traffic.json", "commit": d.get("_commit"), "same_kernel_sources_as_this_build": same,
                "total_over_algorithmic_bytes": amp["total_over_algorithmic"] if amp else None,
                "stale_bytes_per_launch": None if same else fam["bytes_per_launch"]}
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32, help="GLOBAL batch (worker.py:35 splits it over the ranks)")
    ap.add_argument("--dtype", choices=["bf16", "f32", "fp8"], default="bf16",
                    help="fp8 = BASELINE configs[4]: MX-fp8 operands on the eligible convolutions, bf16 feature maps")
    ap.add_argument("--epoch-type", choices=["r1", "odd", "even", "cycle"], default="r1")
    ap.add_argument("--freezeD-layer", type=int, default=-1,
                    help="BASELINE config 4: freeze the first discriminator layers (main.py --freezeD_layer, with freezeD_start 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-cycle", action="store_true", help="time a full warmed 8-iteration cycle of the CPU oracle at batch 4 (~4 minutes) instead of one iteration per type")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--launch-table", default="", help="write the per-launch table (HIP events, conv geometry tags) of one more iteration here")
    ap.add_argument("--h2d", action="store_true", help="PCIe-inclusive variant: the three views of every iteration come from pinned host memory "
                    "(async copy one iteration ahead, worker.SyntheticHostTriples) instead of being resident in HBM; never the headline value")
    ap.add_argument("--no-cycle", action="store_true", help="skip the extra 8-iteration cycle (epoch 8..15) reported beside the R1 line")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # LCGAN_BENCH_DEVICE=cpu exists for tests/test_bench_gloo.py only: it walks this file's world > 1 control flow (collective counts,
    # lock-step of the ranks) on the host with the TESTS' kernel emulation installed.  The product has no CPU kernels: without that
    # injection the first kernel call raises.
    on_gpu = os.environ.get("LCGAN_BENCH_DEVICE", "cuda") != "cpu"
    if on_gpu:
        ndev = torch.cuda.device_count()
        dev_index = local_rank % max(ndev, 1)              # (ranks > devices only happens in the single-GPU dry run below)
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    else:
        dev = torch.device("cpu")

    def dsync():
        if on_gpu:
            torch.cuda.synchronize()
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("LCGAN_DIST_BACKEND", "nccl")      # "nccl" == RCCL; "gloo" lets two ranks share one GPU for a dry run
        if backend == "nccl" and on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from lcgan_amd import config, kernels, loader, worker
    from lcgan_amd.config import default_args as make_args
    config.set_feature_dtype(torch.float32 if a.dtype == "f32" else torch.bfloat16)
    config.set_conv_operands("fp8" if a.dtype == "fp8" else "bf16")
    assert kernels.backend_name() == "hip" or not on_gpu
    extra = dict(freezeD_start=0, freezeD_layer=a.freezeD_layer) if a.freezeD_layer >= 0 else {}
    args = make_args(a.res, a.batch, **extra)
    if a.h2d:
        args.dataset_path = "synthetic-host"
    torch.manual_seed(0)                                   # identical reference-style init on every rank
    w = worker.WORKER(args, local_rank, world, device=dev)
    torch.manual_seed(1 + rank)                            # different latents per rank

    def epoch_of(i):
        return {"r1": 1 + 8 * i, "odd": 3 + 8 * i, "even": 8 * i, "cycle": i}[a.epoch_type]

    def sync():
        if world > 1:
            dist.barrier()
        dsync()

    for i in range(a.warmup):
        loader.train_iteration(w, args, epoch_of(i))
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loader.train_iteration(w, args, epoch_of(a.warmup + i))
    w.flush()                                              # postponed all-reduce wait + Adam of the last D step (N > 1)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    images = a.batch * a.steps
    value = images / dt
    fl_step = sum(flops_per_image(a.res, epoch_of(a.warmup + i)) for i in range(a.steps)) / a.steps * a.batch
    out = {
        "metric": "images/sec (G+D step)", "value": value, "unit": "images/sec", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic, views copied from pinned host memory every iteration" if a.h2d else "synthetic",
        "config": {"workload": f"LC-GAN G+D iteration ({a.epoch_type}: train_generator + ema + train_discriminator"
                               f"{' with R1' if a.epoch_type == 'r1' else ''}), {a.res}x{a.res}, global batch {a.batch}"
                               f"{f', freezeD_layer {a.freezeD_layer}' if a.freezeD_layer >= 0 else ''}",
                   "global_batch": a.batch, "resolution": a.res, "parallelism": f"dp{world}",
                   "algorithmic_tflop_per_step": fl_step / 1e12},
        # (with frozen discriminator layers part of the backward is skipped: the full-step FLOP formula does not apply)
        "step_mfma_frac": fl_step / (dt / a.steps) / (world * PEAK_BF16_DENSE) if a.freezeD_layer < 0 else None,
    }

    if not a.no_cycle and a.epoch_type == "r1":
        # the 8-iteration cycle mean (4 even + 1 R1 + 3 odd iterations) beside the headline R1 iteration
        for e in range(8, 16):
            loader.train_iteration(w, args, e)                  # warm-up cycle: the even iterations' shapes / prepared weights
        w.flush()
        sync()
        tc = time.perf_counter()
        for e in range(16, 24):
            loader.train_iteration(w, args, e)
        w.flush()
        sync()
        dtc = time.perf_counter() - tc
        if world > 1:
            t = torch.tensor([dtc], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtc = float(t.item())
        fl_cycle = sum(flops_per_image(a.res, e) for e in range(16, 24)) * a.batch
        out["cycle"] = {"value": 8 * a.batch / dtc, "ms_per_step_mean": dtc / 8 * 1e3, "iterations": "epoch 16..23 (4 even, 1 odd+R1, 3 odd)",
                        "step_mfma_frac": fl_cycle / dtc / (world * PEAK_BF16_DENSE) if a.freezeD_layer < 0 else None}

    if rank == 0 and not a.no_roofline:
        # dominant kernel (implicit-GEMM conv): HIP events around every launch, on the launch stream, over one more iteration
        K = kernels.K
        K.prof_enable(True)
        loader.train_iteration(w, args, epoch_of(a.warmup + a.steps))
        dsync()
        prof = K.prof_collect()
        K.prof_enable(False)
        none = {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "count": 0}
        ig, wg = prof.get("conv_igemm", none), prof.get("conv_wgrad", none)
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        # `traffic` (memory-side bytes per launch) needs PMC passes under rocprofv3 and cannot be measured inside this run: the separately
        # profiled figure of the same command (scripts/pmc_traffic.py -> profiles/r04_pmc_traffic.json, FETCH_SIZE x2 on gfx950) is quoted
        # while the running build's kernel sources match the profiled build's; only for the configuration it was profiled on (256 x 256, batch 32, bf16)
        tr = committed_traffic() if (a.res == 256 and a.batch == 32 and a.dtype == "bf16" and world == 1) else None
        out["roofline"] = {"bound": "mfma", "kernel": "conv_halo_kernel + conv_igemm_kernel (forward / data-gradient convolutions)",
                           "achieved": ach, "peak": PEAK_BF16_DENSE / 1e12,
                           "unit": "TFLOP/s", "frac": ach / (PEAK_BF16_DENSE / 1e12), "traffic": tr["bytes_per_launch"] if tr else None,
                           "traffic_source": tr,
                           "launches": ig["count"], "avg_launch_ms": ig["ms"] / max(ig["count"], 1),
                           "flops_per_launch": ig["flops"] / max(ig["count"], 1)}
        out["kernel_ms"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        out["wgrad_tflops"] = wg["flops"] / (wg["ms"] * 1e-3) / 1e12 if wg["ms"] > 0 else 0.0
        # HBM roofline of the memory-bound kernel families (K3-K6, K17, K18): algorithmic bytes / HIP-event time
        hbm = {k: {"GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "frac": v["bytes"] / (v["ms"] * 1e-3) / PEAK_HBM, "ms": round(v["ms"], 3)}
               for k, v in prof.items() if k in ("act_bwd", "stencil", "warp_fwd", "warp_bwd", "optim", "rgb") and v["ms"] > 0 and v["bytes"] > 0}
        if hbm:
            top = max(hbm, key=lambda k: hbm[k]["ms"])
            out["roofline_hbm"] = {"bound": "hbm", "kernel": top, "achieved": hbm[top]["GBps"], "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                                   "frac": hbm[top]["frac"], "traffic": None, "families": hbm}
        if a.launch_table:                                  # per-launch in-iteration table: shape -> us -> TFLOP/s (profiles/*_launch_table.csv)
            K.prof_enable(True)
            loader.train_iteration(w, args, epoch_of(a.warmup + a.steps + 1))
            dsync()
            K.prof_dump(a.launch_table)
            K.prof_enable(False)
    elif world > 1 and not a.no_roofline:
        # keep the ranks in lock-step: every iteration rank 0 runs above carries gradient all-reduces that need their peers
        loader.train_iteration(w, args, epoch_of(a.warmup + a.steps))
        if a.launch_table:
            loader.train_iteration(w, args, epoch_of(a.warmup + a.steps + 1))
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.res, full_cycle=a.cpu_baseline_cycle)
    if world > 1:
        w.flush()                                          # the roofline iteration's postponed all-reduce wait + Adam
        dsync()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
