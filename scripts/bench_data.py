"""Throughput of the folder data source (lcgan_amd/data.py:FolderTriples): images/s delivered to the training thread from a folder of
JPEGs, with the decode thread pool one batch ahead.  python scripts/bench_data.py [res] [batch] [workers] [source size]
On a GPU box the views are made by the HIP kernel and a busy-wait stands in for the training step; without a GPU the CPU emulation of
the view kernel is excluded from the timing (decode + resize + parameter draws only)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from lcgan_amd import data

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
srcsize = int(sys.argv[4]) if len(sys.argv) > 4 else res
root = tempfile.mkdtemp(prefix="lcgan_data_")
os.makedirs(os.path.join(root, "train", "c"))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, size=(srcsize // 8, srcsize // 8, 3), dtype=np.uint8)
for i in range(4 * batch):                                   # smooth-ish images (JPEG of pure noise is unrepresentative)
    a = np.asarray(Image.fromarray(np.roll(base, i, axis=0)).resize((srcsize, srcsize), Image.BICUBIC))
    Image.fromarray(a).save(os.path.join(root, "train", "c", f"{i:05d}.jpg"), quality=90)
gpu = torch.cuda.is_available()
if not gpu:
    class _NoViews:
        def make_views(self, src, par):
            return src, src, src
src = data.FolderTriples(root, res, batch, "cuda:0" if gpu else "cpu", workers=workers)
if not gpu:
    src.K = _NoViews()
for _ in range(2):
    src.next()
if gpu:
    torch.cuda.synchronize()
n = 12
t0 = time.perf_counter()
wait = 0.0
for _ in range(n):
    t1 = time.perf_counter()
    out = src.next()
    wait += time.perf_counter() - t1
    if gpu:
        torch.cuda.synchronize()
    time.sleep(0.08)                                          # the training step the decode overlaps with (82 ms at batch 32)
dt = time.perf_counter() - t0
print(f"res {res} (source {srcsize}) batch {batch} workers {workers} gpu {gpu}: {n * batch / dt:.0f} img/s delivered with an 80 ms step "
      f"between batches; training thread blocked in next() {1e3 * wait / n:.1f} ms per batch; decode-only rate {n * batch / max(dt - 0.08 * n, 1e-9):.0f} img/s")
src.close()
