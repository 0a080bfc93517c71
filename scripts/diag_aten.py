"""GPU diagnostic: which aten ops (torch glue) still run in one R1 iteration, attributed to the innermost repo frame."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from lcgan_amd import config, loader, worker
from lcgan_amd.config import default_args as make_args
config.set_feature_dtype(torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = make_args(256, B)
torch.manual_seed(0)
w = worker.WORKER(args, 0, 1)
for ep in (1, 9):
    loader.train_iteration(w, args, ep)
torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cnt = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.empty", "aten.as_strided", "aten._unsafe_view", "aten.t.", "aten.slice", "aten.select", "aten.unsqueeze",
        "aten.squeeze", "aten.permute", "aten.transpose", "aten.expand", "aten.alias", "aten.reshape", "aten._local_scalar", "aten.is_", "aten.sym_",
        "aten.flatten", "aten.unbind", "aten.split", "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.set_")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "diag_aten" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    loader.train_iteration(w, args, 17)
torch.cuda.synchronize()
tot = sum(cnt.values())
print("aten ops that do work:", tot)
for (name, site), c in cnt.most_common(45):
    print(f"{c:4d}  {name:32s} {site}")
