"""Time the one-workgroup Householder QR (csrc/small.hip) on the GPU: nb matrices of n x n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import kernels
H = kernels.HipKernels()
for nb, n in ((1, 64), (2, 64), (8, 64), (1, 32)):
    A = torch.tanh(torch.randn(nb, n, n, device="cuda"))
    for _ in range(5):
        H.qr(A)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        H.qr(A)
    e1.record(); torch.cuda.synchronize()
    Q, R = H.qr(A)
    err = (Q @ R - A).abs().max().item()
    print(f"qr nb={nb} n={n}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us/call  |QR-A|max={err:.2e}")
