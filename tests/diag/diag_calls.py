"""GPU diagnostic: run a G step with every kernel call executed by BOTH the HIP library and the CPU emulation on the same
inputs; print the calls whose outputs differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lcgan_amd.kernels as KM
from lcgan_amd import config
from lcgan_amd.kernels import HipKernels, PreparedWeight
from oracle.hip_emulation import EmulatedKernels, EmuWeight
from tests.helpers import install_backend, FixedFeed, seeded_worker

H, E = HipKernels(), EmulatedKernels()
LOG = []


def to_cpu(a, wmap):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().clone()
    if isinstance(a, PreparedWeight):
        return wmap[id(a)]
    return a


def cmp(name, h, e, idx):
    if isinstance(h, torch.Tensor):
        hf, ef = h.detach().float().cpu(), e.detach().float()
        sc = ef.abs().max().clamp_min(1e-30)
        mx = float((hf - ef).abs().max() / sc)
        l2 = float((hf - ef).norm() / ef.norm().clamp_min(1e-30))
        LOG.append((mx, l2, name, idx, tuple(h.shape)))


class Checked:
    name = "hip"

    def __init__(self):
        self.wmap = {}
        self.n = 0

    def __getattr__(self, item):
        hf, ef = getattr(H, item), getattr(E, item)

        def call(*args, **kw):
            self.n += 1
            cargs = [to_cpu(a, self.wmap) for a in args]
            ckw = {k: to_cpu(v, self.wmap) for k, v in kw.items()}
            out_h = hf(*args, **kw)
            out_e = ef(*cargs, **ckw)
            if item == "prep_weight":
                self.wmap[id(out_h[0])] = out_e[0]
                self._keep = getattr(self, "_keep", []) + [out_h[0]]
                if out_h[1] is not None:
                    cmp(item + ".wsq", out_h[1], out_e[1], 1)
                return out_h
            outs_h = out_h if isinstance(out_h, tuple) else (out_h,)
            outs_e = out_e if isinstance(out_e, tuple) else (out_e,)
            for i, (a, b) in enumerate(zip(outs_h, outs_e)):
                if a is not None:
                    cmp(f"#{self.n} {item}", a, b, i)
            if item in ("demod_bwd",):          # in-place gs (args[4])
                cmp(f"#{self.n} {item}.gs", args[4], cargs[4], 9)
            if item == "avg_latent":
                cmp(f"#{self.n} {item}.avg", args[1], cargs[1], 9)
            return out_h
        return call


res, B = int(sys.argv[1]), int(sys.argv[2])
config.set_feature_dtype(torch.float32)
install_backend(Checked())
w = seeded_worker(res, B, "cuda:0")
FixedFeed(w, B, res, "cuda:0")
w.g_optimizer.step = lambda: None
w.requires_grad(w.generator, True), w.requires_grad(w.discriminator, False)
w.train_generator(1)
print("calls:", len(LOG))
for mx, l2, name, idx, shape in sorted(LOG, reverse=True)[:40]:
    print(f"{mx:.3e} {l2:.3e} {name}[{idx}] {shape}")
