"""Host-side cost of one G+D iteration at a small batch (where the step is launch-bound): cProfile over a few iterations."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import config, worker
from lcgan_amd.config import default_args
config.set_feature_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = worker.WORKER(default_args(256, B), 0, 1, device=dev)
for ep in (0, 1, 9, 17):
    w.train_generator(ep); w.train_discriminator(ep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for ep in (17, 25, 33, 41, 49):
    w.train_generator(ep); w.train_discriminator(ep)
t1 = time.perf_counter()                       # host time to ENQUEUE five iterations
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: enqueue {1e3 * (t1 - t0) / 5:.2f} ms/iter, drained after {1e3 * (t2 - t0) / 5:.2f} ms/iter")
pr = cProfile.Profile()
pr.enable()
for ep in (17, 25, 33):
    w.train_generator(ep); w.train_discriminator(ep)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
