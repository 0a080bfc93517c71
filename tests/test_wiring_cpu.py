"""Host-side logic on CPU: the autograd wiring (ops.py), the nn.Module surface (custom_layers.py, cnn.py), the losses and the
worker's step sequencing are driven with a CPU emulation of the kernel interface (oracle/hip_emulation.py, installed through
tests/helpers.py:install_backend) and compared with the golden vectors captured from the reference.
The HIP kernels themselves are checked on the GPU (test_kernels_gpu.py, test_parity_gpu.py)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import lcgan_ref as O
from oracle.hip_emulation import EmulatedKernels
from oracle.weights import seeded_state, seeded_tensor
from tests.helpers import install_backend, GOLD, FixedFeed, check_grads_vs_golden, seeded_worker

TOL = 1e-3        # north_star tolerance (relative, fp32)
# Gradients are compared through L1 / L2 norms and seeded random projections (tests/helpers.py): elementwise maxima are
# ill-posed because a leaky-ReLU pre-activation within ~1e-7 of zero flips its mask under any change of summation order
# (measured: one such flip of 131 072 elements moves single weight-gradient entries by 2.6e-2).
# Even iterations add the contrastive loss exp(sim / tau), tau = 0.05: rounding of the embeddings is amplified 20x and
# few-element gradients (the 2-value flow bias) are sums with heavy cancellation; two fp32-correct implementations
# (reference vs oracle) already differ by ~1e-3 there, so even-iteration gradients are held to 3e-3.
TOL_EVEN_GRADS = 3e-3


@pytest.fixture(autouse=True)
def emulated_backend():
    import lcgan_amd.kernels as KM
    from lcgan_amd import config
    install_backend(EmulatedKernels())
    with config.feature_dtype_as(torch.float32):
        yield
    install_backend(None)


@pytest.fixture(scope="module")
def S():
    return np.load(os.path.join(GOLD, "step_r32.npz"))


def rel(a, b):
    a, b = torch.as_tensor(np.asarray(a), dtype=torch.float64), torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_state_dict_layout_matches_reference(S):
    from tests.helpers import make_args
    from lcgan_amd import cnn
    G, D = cnn.Generator(make_args(32)), cnn.Discriminator(make_args(32))
    assert sorted(G.state_dict()) == list(S["g_keys"]) and sorted(D.state_dict()) == list(S["d_keys"])
    assert [str(tuple(G.state_dict()[k].shape)) for k in sorted(G.state_dict())] == list(S["g_shapes"])
    assert [str(tuple(D.state_dict()[k].shape)) for k in sorted(D.state_dict())] == list(S["d_shapes"])
    # freezeD relies on the children order [conv1x1, LeakyReLU, block, ...] (worker.py:128-131)
    names = [type(m).__name__ for m in D.shared_model.children()]
    assert names[:3] == ["EqualizedConv2d", "LeakyReLU", "DiscriminatorBlock"]


@pytest.mark.parametrize("epoch", [0, 1])
def test_train_generator_matches_reference(S, epoch):
    res, B = int(S["res"]), int(S["B"])
    w = seeded_worker(res, B, "cpu")
    FixedFeed(w, B, res, "cpu")
    w.g_optimizer.step = lambda: None                       # inspect the gradients before Adam consumes them
    w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
    g_loss = w.train_generator(epoch)
    assert rel(g_loss, S[f"g{epoch}/loss"]) <= TOL
    check_grads_vs_golden(S, f"g{epoch}", w.generator.module.named_parameters(), TOL if epoch % 2 else TOL_EVEN_GRADS)
    assert rel(w.generator.module.avg_latent1, S[f"g{epoch}/avg_latent1"]) <= TOL
    assert rel(w.generator.module.avg_latent2, S[f"g{epoch}/avg_latent2"]) <= TOL
    assert all(p.grad is None for p in w.discriminator.parameters())


@pytest.mark.parametrize("epoch,frozen", [(0, 0), (1, 0), (3, 0), (1, 2)])
def test_train_discriminator_matches_reference(S, epoch, frozen):
    res, B = int(S["res"]), int(S["B"])
    tag = f"d{epoch}" + (f"_freeze{frozen}" if frozen else "")
    w = seeded_worker(res, B, "cpu")
    FixedFeed(w, B, res, "cpu")
    w.d_optimizer.step = lambda: None
    w.requires_grad(w.generator, False), w.requires_grad(w.discriminator, True)
    if frozen:
        w.freeze_discriminator(frozen)
    d_loss = w.train_discriminator(epoch)
    assert rel(d_loss, S[f"{tag}/loss"]) <= TOL
    check_grads_vs_golden(S, tag, w.discriminator.module.named_parameters(), TOL if epoch % 2 else TOL_EVEN_GRADS)
    none = sorted(k for k, p in w.discriminator.module.named_parameters() if p.grad is None)
    assert none == sorted(k for k in S[f"{tag}/grad_none"] if k)


@pytest.mark.parametrize("which,epoch", [("g", 0), ("d", 0), ("d", 3)])
def test_batched_passes_equal_separate_passes(S, which, epoch):
    """worker.py evaluates G three times and D three / four times on an even iteration and D twice on an odd one (worker.py:163-169,
    194-200).  The product runs those calls as ONE batch (worker.WORKER: config.batched_passes) with the minibatch-stddev statistic
    and the avg-latent updates kept per call (n_sub).  Both forms must give the reference's numbers (the golden tests above run the
    batched form, this one runs the separate form against the same goldens) and agree with each other far below the golden tolerance."""
    from lcgan_amd import config
    res, B = int(S["res"]), int(S["B"])
    out = {}
    for batched in (True, False):
        config.set_batched_passes(batched)
        try:
            w = seeded_worker(res, B, "cpu")
            FixedFeed(w, B, res, "cpu")
            w.g_optimizer.step = w.d_optimizer.step = lambda: None
            if which == "g":
                w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
                lossv, net = w.train_generator(epoch), w.generator.module
            else:
                w.requires_grad(w.generator, False), w.requires_grad(w.discriminator, True)
                lossv, net = w.train_discriminator(epoch), w.discriminator.module
            out[batched] = (float(lossv), {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()},
                            w.generator.module.avg_latent1.clone(), w.generator.module.avg_latent2.clone())
            if not batched:                                 # the separate form against the reference's goldens too
                assert rel(lossv, S[f"{which}{epoch}/loss"]) <= TOL
                check_grads_vs_golden(S, f"{which}{epoch}", net.named_parameters(), TOL if epoch % 2 else TOL_EVEN_GRADS)
        finally:
            config.set_batched_passes(True)
    (la, ga, a1, a2), (lb, gb, b1, b2) = out[True], out[False]
    assert abs(la - lb) <= 1e-5 * abs(lb)
    assert torch.allclose(a1, b1, rtol=1e-5, atol=1e-7) and torch.allclose(a2, b2, rtol=1e-5, atol=1e-7)      # avg-latent: one update per call, in order
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None), k
        if ga[k] is not None:
            err = float((ga[k] - gb[k]).norm() / gb[k].norm().clamp_min(1e-30))
            assert err <= (TOL if epoch % 2 else TOL_EVEN_GRADS), (k, err)   # (summation order of the batch reductions; the contrastive terms amplify fp32 rounding 20x, see TOL_EVEN_GRADS)


def test_forward_truncation_and_deepcopy():
    """w_psi > 0 branch (cnn.py:99-101) and that the DataParallel-wrapped generator survives deepcopy (worker.py:40)."""
    w = seeded_worker(16, 2, "cpu")
    z1, z2 = seeded_tensor((2, 64), 1), seeded_tensor((2, 64), 2)
    GP = {k: v.clone() for k, v in w.generator.module.state_dict().items()}
    with torch.no_grad():
        got = w.generator_ema(z1, z2, 0.7)
        ref = O.generator_forward({k: v.clone() for k, v in w.generator_ema.module.state_dict().items()}, z1, z2, 16, w_psi=0.7)
    assert rel(got, ref) <= TOL
    assert list(w.generator_ema.state_dict())[0].startswith("module.")
    assert all(torch.equal(a, b) for a, b in zip(GP.values(), w.generator.module.state_dict().values()))


def test_adam_matches_torch_and_skips_unused():
    from lcgan_amd.optim import Adam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s)) for s in ((5, 3), (70000,), (1,))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, topt = Adam(ps, lr=0.002, betas=(0.0, 0.99)), torch.optim.Adam(ref, lr=0.002, betas=(0.0, 0.99), eps=1e-8)
    for it in range(4):
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i == 2 and it % 2 == 0:                       # parameter 2 is unused on even iterations
                p.grad, r.grad = None, None
            else:
                g = torch.randn_like(p)
                p.grad, r.grad = g.clone(), g.clone()
        opt.step(), topt.step()
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-7)
    assert opt.steps == [4, 4, 2]


def test_ema_matches_reference_semantics():
    from lcgan_amd.ema import Ema
    w = seeded_worker(16, 2, "cpu")
    src, tgt = w.generator, w.generator_ema
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.1)
        src.module.avg_latent1.add_(1.0)
    before = {k: v.clone() for k, v in tgt.state_dict().items()}
    e = Ema(src, copy.deepcopy(tgt), decay=0.9, start_iter=3)      # ctor copies source -> target (ema.py:13-17)
    assert all(torch.equal(a, b) for a, b in zip(e.target.state_dict().values(), src.state_dict().values()))
    e.target.load_state_dict(before)
    e.update(5)
    for k, v in e.target.state_dict().items():
        s = src.state_dict()[k]
        assert torch.allclose(v, s + 0.9 * (before[k] - s), rtol=1e-6, atol=1e-7), k
    e.update(1)                                                    # iter < start_iter: decay 0 -> exact copy (ema.py:20-23)
    assert all(torch.equal(a, b) for a, b in zip(e.target.state_dict().values(), src.state_dict().values()))


def test_checkpoint_round_trip_and_reference_layout(S, tmp_path):
    """worker.py:219-253, loader.py:35-42: (1) save_model -> load_model into a fresh WORKER restores G, G_ema, D (same outputs) and
    the Adam state; (2) the files carry the `module.`-prefixed keys of the reference's DDP-wrapped modules (key dump captured from
    the reference: g_keys / d_keys of step_r32.npz); (3) a checkpoint WRITTEN in the reference's layout (those keys, no optimizer
    file) loads."""
    from lcgan_amd import loader
    from tests.helpers import make_args
    res, B = 16, 2
    args = dict(model_name=str(tmp_path / "run"), save_dir="model")
    os.makedirs(tmp_path / "run" / "model")
    torch.manual_seed(3)
    w = seeded_worker(res, B, "cpu", **args)
    FixedFeed(w, B, res, "cpu")
    for epoch in (0, 1):
        loader.train_iteration(w, w.args, epoch)                    # moves G, D, G_ema and both Adam states
    w.save_model()
    files = sorted(os.listdir(tmp_path / "run" / "model"))
    assert files == ["disc_model.ckpt", "gen_ema_model.ckpt", "gen_model.ckpt", "optim_state.ckpt"]
    sd_g, sd_d = torch.load(tmp_path / "run" / "model" / "gen_model.ckpt"), torch.load(tmp_path / "run" / "model" / "disc_model.ckpt")
    g_ref = {"module." + k for k in O.g_param_shapes(res)}          # the oracle's inventory == the reference's dump (test_oracle_golden)
    assert set(sd_g) == g_ref and set(sd_d) == {"module." + k for k in O.d_param_shapes(res)}
    assert sorted(k[len("module."):] for k in torch.load(tmp_path / "run" / "model" / "gen_ema_model.ckpt")) == sorted(O.g_param_shapes(res))

    torch.manual_seed(4)
    w2 = worker_mod().WORKER(make_args(res, B, **args), 0, 1, device="cpu")   # different random init
    w2.load_model()
    for a, b in ((w.generator, w2.generator), (w.generator_ema, w2.generator_ema), (w.discriminator, w2.discriminator)):
        for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(va, vb), k
    assert w2.g_optimizer.steps == w.g_optimizer.steps and w2.d_optimizer.steps == w.d_optimizer.steps
    assert all(torch.equal(a, b) for a, b in zip(w.d_optimizer.exp_avg_sq, w2.d_optimizer.exp_avg_sq))
    z1, z2 = seeded_tensor((B, 64), 1), seeded_tensor((B, 64), 2)
    with torch.no_grad():
        assert torch.equal(w.generate(z1, z2, 0.7), w2.generate(z1, z2, 0.7))
    # resumed training continues identically (Adam moments included)
    FixedFeed(w2, B, res, "cpu")
    FixedFeed(w, B, res, "cpu")
    la, lb = loader.train_iteration(w, w.args, 3), loader.train_iteration(w2, w2.args, 3)
    assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1])
    for (k, va), (_, vb) in zip(w.discriminator.state_dict().items(), w2.discriminator.state_dict().items()):
        assert torch.equal(va, vb), k

    # (3) a reference-written checkpoint: state_dicts keyed exactly like the reference's dump, no optimizer file
    ref_dir = tmp_path / "ref" / "model"
    os.makedirs(ref_dir)
    GP, DP = seeded_state(O.g_param_shapes(res), 77), seeded_state(O.d_param_shapes(res), 78)
    torch.save({"module." + k: v for k, v in GP.items()}, ref_dir / "gen_model.ckpt")
    torch.save({"module." + k: v for k, v in GP.items()}, ref_dir / "gen_ema_model.ckpt")
    torch.save({"module." + k: v for k, v in DP.items()}, ref_dir / "disc_model.ckpt")
    w3 = worker_mod().WORKER(make_args(res, B, model_name=str(tmp_path / "ref"), save_dir="model"), 0, 1, device="cpu")
    w3.load_model()
    assert all(torch.equal(v, GP[k]) for k, v in w3.generator.module.state_dict().items())
    assert all(torch.equal(v, DP[k]) for k, v in w3.discriminator.module.state_dict().items())
    with torch.no_grad():
        img = w3.generate(z1, z2, 0.7)
    assert rel(img, O.generator_forward({k: v.clone() for k, v in GP.items()}, z1, z2, res, w_psi=0.7)) <= TOL


def worker_mod():
    from lcgan_amd import worker
    return worker


def test_fake_image_generation_phase(tmp_path):
    """loader.py:95-99 + worker.py:427-441: load the checkpoints, write num_fakes column images of the EMA generator's output."""
    from PIL import Image
    from lcgan_amd import loader
    from tests.helpers import make_args
    res, B = 16, 2
    run = tmp_path / "run"
    os.makedirs(run / "model")
    w = seeded_worker(res, B, "cpu", model_name=str(run), save_dir="model")
    w.save_model()
    args = make_args(res, B, model_name=str(run), save_dir="model", phase="fake_image_generation", num_fakes=3, w_psi=0.7)
    torch.manual_seed(11)
    gw = worker_mod().WORKER(args, 0, 1, device="cpu")
    gw.load_model()
    gw.fake_image_generation(num_images=args.num_fakes)
    names = sorted(os.listdir(run / "fakes"))
    assert names == ["0000_images.jpg", "0001_images.jpg", "0002_images.jpg"]
    assert Image.open(run / "fakes" / names[0]).size == (res, B * res)
    torch.manual_seed(11)                                            # the same draws -> the first file's pixels
    ref = ((w.generate(torch.randn(B, 64), torch.randn(B, 64), 0.7) + 1) / 2).clamp(0, 1)
    got = torch.from_numpy(np.array(Image.open(run / "fakes" / names[0]))).float() / 255
    assert float((got - ref.permute(0, 2, 3, 1).reshape(B * res, res, 3)).abs().mean()) < 0.05      # JPEG is lossy


def test_product_has_no_cpu_fallback():
    """Without the test hook the product path must refuse CPU tensors loudly (no silent eager fallback)."""
    import lcgan_amd.kernels as KM
    install_backend(None)
    try:
        w = None
        with pytest.raises((RuntimeError, AssertionError)):
            from lcgan_amd import worker
            from tests.helpers import make_args
            w = worker.WORKER(make_args(16, 2), 0, 1)               # device cuda:0 does not exist here
        with pytest.raises((RuntimeError, AssertionError)):
            from lcgan_amd import ops
            ops.Box3Fn.apply(torch.zeros(1, 4, 4, 8))
    finally:
        install_backend(EmulatedKernels())


def test_qr_backward_formula_matches_torch():
    """ops.QrQFn's hand-derived backward (gA = Q tril(G - G^T, -1) R^-T) against autograd through torch.linalg.qr."""
    from lcgan_amd import ops
    A = torch.tanh(seeded_tensor((24, 24), 5)).requires_grad_(True)
    gQ = seeded_tensor((24, 24), 6)
    (torch.linalg.qr(A, mode="reduced")[0] * gQ).sum().backward()
    ref = A.grad.clone()
    A.grad = None
    (ops.QrQFn.apply(A) * gQ).sum().backward()
    assert rel(A.grad, ref) <= 1e-4


def test_qr_batched_backward_matches_torch():
    """the batched form (both mapping networks in one launch) against autograd through torch.linalg.qr"""
    from lcgan_amd import ops
    A = torch.tanh(seeded_tensor((2, 16, 16), 7)).requires_grad_(True)
    gQ = seeded_tensor((2, 16, 16), 8)
    (torch.linalg.qr(A, mode="reduced")[0] * gQ).sum().backward()
    ref = A.grad.clone()
    A.grad = None
    (ops.QrQFn.apply(A) * gQ).sum().backward()
    assert rel(A.grad, ref) <= 1e-4


def test_prepared_weights_are_rebuilt_in_groups_after_an_optimiser_step():
    """ops._prep: the first iteration prepares weights one by one and records the recipes; after Adam invalidated a network's
    parameters the first miss rebuilds ALL its recorded variants in one grouped call, and the results are the same tensors a
    single preparation gives."""
    import lcgan_amd.kernels as KM
    from lcgan_amd import loader, ops
    from tests.helpers import make_args
    calls = {"single": 0, "group": 0, "jobs": 0}
    be = KM._Lazy._impl
    single, group = be.prep_weight, be.prep_weight_group

    def count_single(*a, **k):
        calls["single"] += 1
        return single(*a, **k)

    def count_group(jobs):
        calls["group"] += 1
        calls["jobs"] += len(jobs)
        return [single(*j) for j in jobs]

    be.prep_weight, be.prep_weight_group = count_single, count_group
    try:
        w = seeded_worker(32, 4, torch.device("cpu"))
        args = make_args(32, 4)
        loader.train_iteration(w, args, 1)
        first = dict(calls)
        # nothing recorded at first: one by one (only the generator, already stepped by Adam when the D step runs it again,
        # can take the grouped path inside the first iteration)
        assert first["single"] > 20 and first["group"] <= 1
        loader.train_iteration(w, args, 3)
        assert calls["group"] - first["group"] >= 2 and calls["jobs"] > 20    # G's and D's variants, each rebuilt by one grouped call
        assert calls["single"] - first["single"] < first["single"] // 4    # hardly any single preparation left
        # the cache holds what a fresh single preparation returns
        conv = w.discriminator.module.shared_model[2].conv0.weight
        pw, _ = ops._prep(conv.weight, conv.c, False, True)
        ref, _ = single(conv.weight, conv.c, False, True)
        assert torch.equal(pw.P4, ref.P4)
    finally:
        be.prep_weight, be.prep_weight_group = single, group


def test_fid_statistics_vs_reference():
    """lcgan_amd.fid.calc_fid against values computed by the reference's eval/fid.py (tests/golden/fid.npz, oracle/make_golden.py)."""
    from lcgan_amd import fid
    Fz = np.load(os.path.join(GOLD, "fid.npz"))
    for name in ("a", "b", "c"):
        n, f, shift = int(Fz[f"{name}/n"]), int(Fz[f"{name}/f"]), float(Fz[f"{name}/shift"])
        x = seeded_tensor((n, f), 4000 + n).double().numpy()
        y = (seeded_tensor((n, f), 4100 + n).double() * 1.3 + shift).numpy()
        ref = float(Fz[f"{name}/fid"])
        if np.isnan(ref):
            with pytest.raises(ValueError):
                fid.calc_fid(*fid.feature_statistics(x), *fid.feature_statistics(y))
        else:
            got = fid.calc_fid(*fid.feature_statistics(x), *fid.feature_statistics(y))
            assert abs(got - ref) <= 1e-6 * max(abs(ref), 1.0), (name, got, ref)


def test_fid_evaluate_with_a_pluggable_feature_network(tmp_path):
    """WORKER.fid_evaluate (worker.py:381-425) with a stand-in feature extractor (mean-pooled image patches)."""
    w = seeded_worker(16, 4, "cpu")
    extractor = lambda img: torch.nn.functional.avg_pool2d(img.float(), 4).flatten(1)[:, :12]
    v = w.fid_evaluate(extractor, num_batches=6)
    assert np.isfinite(v) and v > 0 and w.best_fid == v


def test_inputs_only_flag_rejects_an_overlapping_backward_pass():
    """ops.inputs_only (loss.cal_derivative's only_inputs=True) is a process global read on autograd's thread: a second backward pass
    that runs while it is set must fail loudly instead of silently dropping its parameter gradients."""
    from lcgan_amd import ops
    from lcgan_amd.kernels import ACT_NONE
    w = torch.nn.Parameter(seeded_tensor((8, 8), 1))
    x1, x2 = seeded_tensor((2, 8), 2).requires_grad_(True), seeded_tensor((2, 8), 3).requires_grad_(True)
    y1 = ops.LinearFn.apply(x1, w, None, 1.0, 0.0, ACT_NONE, 1.0).sum()
    y2 = ops.LinearFn.apply(x2, w, None, 1.0, 0.0, ACT_NONE, 1.0).sum()
    with ops.inputs_only():
        (g1,) = torch.autograd.grad(y1, x1, retain_graph=True)          # pins its graph task; weight gradient skipped
        assert w.grad is None and g1 is not None
        with pytest.raises(RuntimeError, match="overlap"):
            y2.backward()                                               # another graph task under the same flag
    w.grad = None
    y2.backward()                                                       # outside the flag: ordinary backward, weight gradient formed
    assert w.grad is not None


def test_batched_passes_are_bounded_by_addressable_size():
    """the merged pass of n_sub calls must keep its largest feature map under the fast kernels' 32-bit element offsets: the single-GPU
    high-resolution recipes (1024 x 1024 / batch 32: 3.2e9 elements for three generator calls) keep the reference's separate calls"""
    from lcgan_amd import worker
    def can(res, local_batch, n_sub):
        w = worker.WORKER.__new__(worker.WORKER)
        w.args, w.local_batch_size = type("A", (), {"img_resolution": res})(), local_batch
        return w._can_batch(n_sub)
    assert can(256, 32, 4) and can(512, 32, 3) and can(512, 8, 4) and can(1024, 4, 4)
    assert not can(512, 32, 4) and not can(1024, 32, 2) and not can(1024, 32, 3)


def test_bench_quotes_pmc_traffic_only_for_the_profiled_sources(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from profiles/r04_pmc_traffic.json and is null unless the kernel sources of the running build hash
    to what the profiled build hashed to"""
    import json, bench
    from lcgan_amd.build import source_hash
    (tmp_path / "profiles").mkdir()
    fam = {"families": {"conv_halo": {"launches": 246, "bytes_per_launch": 5.0e8}}, "amplification": {"conv fwd/dgrad": {"total_over_algorithmic": 1.26}}, "_commit": "abc"}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    json.dump({**fam, "_srchash": source_hash()}, open(tmp_path / "profiles" / "r04_pmc_traffic.json", "w"))
    t = bench.committed_traffic()
    assert t["bytes_per_launch"] == 5.0e8 and t["same_kernel_sources_as_this_build"] and t["total_over_algorithmic_bytes"] == 1.26
    json.dump({**fam, "_srchash": "stale"}, open(tmp_path / "profiles" / "r04_pmc_traffic.json", "w"))
    t = bench.committed_traffic()
    assert t["bytes_per_launch"] is None and t["stale_bytes_per_launch"] == 5.0e8 and not t["same_kernel_sources_as_this_build"]


def test_host_resident_synthetic_source_hands_out_the_resident_batches():
    """`--dataset_path synthetic-host` (bench.py --h2d): the batches of the resident source, in the same order, through the copy-ahead path"""
    from lcgan_amd.worker import SyntheticHostTriples, SyntheticTriples
    a, b = SyntheticTriples(2, 8, "cpu", seed=7, pool=3), SyntheticHostTriples(2, 8, "cpu", seed=7, pool=3)
    for _ in range(5):
        for x, y in zip(a.next(), b.next()):
            assert torch.equal(x, y)


def test_zero_pool_slices_do_not_share_a_version_counter():
    """kernels._ZeroPool: slices of one slab are saved for backward (demodulation vectors) AND mutated in place by autograd (gradient
    accumulation over two backward passes): an in-place op on one slice must not invalidate a saved neighbour"""
    from lcgan_amd.kernels import _ZeroPool
    pool = _ZeroPool()
    a, b = pool.take((4, 128), torch.device("cpu")), pool.take((4, 128), "cpu")
    assert a.data_ptr() != b.data_ptr() and float(a.abs().sum() + b.abs().sum()) == 0.0
    assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()           # one slab
    x = torch.ones(4, 128, requires_grad=True)
    y = (x * a).sum()                              # `a` is saved for backward
    va = a._version
    b.add_(1.0)                                    # what AccumulateGrad does to a stolen gradient slice
    assert a._version == va
    y.backward()                                   # (raised "modified by an inplace operation" when the slices were views of the slab)
    assert float(b.sum()) == 4 * 128 and float(a.abs().sum()) == 0.0
