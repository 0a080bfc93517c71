"""GPU diagnostic: run-to-run repeatability of the discriminator forward and of d(sum logit)/d(image), per block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lcgan_amd import config
from tests.helpers import seeded_worker
from oracle.weights import seeded_tensor

res, B = 256, 4
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
with config.feature_dtype_as(torch.bfloat16 if mode == "bf16" else torch.float32):
    w = seeded_worker(res, B, dev)
    real = seeded_tensor((B, 3, res, res), 7, "uniform_pm1").to(dev)
    D = w.discriminator
    w.requires_grad(D, True)
    mods = dict(D.module.named_children())

    def run():
        fw, bw = {}, {}
        hooks = []
        def mk(name):
            def hook(m, inp, out):
                t = out[0] if isinstance(out, (tuple, list)) else out
                if torch.is_tensor(t):
                    fw[name] = t.detach().float().clone()
                    if t.requires_grad:
                        t.register_hook(lambda g, n=name: bw.__setitem__(n, g.detach().float().clone()))
            return hook
        for name, m in D.module.named_modules():
            if name and name.count(".") <= 1:
                hooks.append(m.register_forward_hook(mk(name)))
        img = real.clone().requires_grad_(True)
        logit, _, _ = D(img, False)
        (g,) = torch.autograd.grad(logit.sum(), img, create_graph=False)
        for h in hooks:
            h.remove()
        return fw, bw, logit.detach().float(), g.detach().float()

    f1, b1, l1, g1 = run()
    f2, b2, l2, g2 = run()
    def rel(a, b):
        return float((a - b).norm() / (a.norm() + 1e-30))
    print("logit rel diff", rel(l1, l2), " image-grad rel diff", rel(g1, g2), " r1 rel", abs(float(g1.square().sum() - g2.square().sum())) / float(g1.square().sum()))
    print("forward:")
    for k in f1:
        print(f"  {k:40s} {rel(f1[k], f2[k]):.3e}")
    print("backward (grad wrt module output):")
    for k in b1:
        if k in b2:
            print(f"  {k:40s} {rel(b1[k], b2[k]):.3e}")
