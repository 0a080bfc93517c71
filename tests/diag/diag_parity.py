"""GPU diagnostic: per-parameter gradient error of the HIP path (f32 parity mode) against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lcgan_amd import config
from oracle import lcgan_ref as O
from oracle.weights import seeded_state
from tests.helpers import FixedFeed, seeded_worker

res, B = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
config.set_feature_dtype(torch.float32)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30))


for epoch in (1, 0):
    w = seeded_worker(res, B, dev)
    feed = FixedFeed(w, B, res, dev)
    GP, DP = seeded_state(O.g_param_shapes(res), 1001), seeded_state(O.d_param_shapes(res), 1002)
    w.g_optimizer.step = lambda: None
    w.d_optimizer.step = lambda: None
    w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
    g_loss = w.train_generator(epoch)
    w.requires_grad(w.generator, False), w.requires_grad(w.discriminator, True)
    feed.reset()
    d_loss = w.train_discriminator(epoch)
    z = [t.cpu() for t in feed.z]
    real = tuple(t.cpu() for t in feed.real)
    g_ref, g_grads, _, _ = O.g_step(GP, DP, res, epoch, tuple(z))
    d_ref, d_grads, _, parts = O.d_step(GP, DP, res, epoch, (z[0], z[1]), real)
    print(f"epoch {epoch}: g_loss {g_loss:.7f} / {float(g_ref):.7f}   d_loss {d_loss:.7f} / {float(d_ref):.7f}")
    for name, mod, ref in (("G", w.generator.module, g_grads), ("D", w.discriminator.module, d_grads)):
        errs = sorted(((*rel(p.grad, ref[k]), k, p.numel()) for k, p in mod.named_parameters() if k in ref), reverse=True)
        print(f"  {name}: worst max-rel / l2-rel")
        for e in errs[:8]:
            print(f"     {e[0]:.3e} {e[1]:.3e}  {e[2]}  ({e[3]})")
