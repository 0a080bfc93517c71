"""GPU sanity of BASELINE configs 3/4 shapes: one R1 iteration at 512x512 and 1024x1024 (with freezeD_layer=5), tiny batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import config, loader, worker
from lcgan_amd.config import default_args as make_args
config.set_feature_dtype(torch.bfloat16)
for res, B, kw in ((512, 2, {}), (1024, 2, dict(freezeD_start=0, freezeD_layer=5, g_lr=0.001, d_lr=0.001))):
    torch.manual_seed(0)
    args = make_args(res, B, **kw)
    w = worker.WORKER(args, 0, 1)
    t0 = time.perf_counter()
    for ep in (1, 0, 9):
        gl, dl = loader.train_iteration(w, args, ep)
        print(res, "epoch", ep, "g_loss %.4f d_loss %.4f" % (float(gl), float(dl)), flush=True)
    torch.cuda.synchronize()
    frozen = [k for k, p in w.discriminator.module.named_parameters() if not p.requires_grad]
    print(res, "ok %.1fs" % (time.perf_counter() - t0), "frozen params:", len(frozen), "mem GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del w
    torch.cuda.empty_cache()
