"""Which torch-side ops launch the small fill / copy / add kernels of one R1 iteration (GPU, torch.profiler with stacks)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from lcgan_amd import config, worker
from lcgan_amd.config import default_args
config.set_feature_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
w = worker.WORKER(default_args(256, int(sys.argv[1]) if len(sys.argv) > 1 else 4), 0, 1, device=dev)
for ep in (0, 1, 9):
    w.train_generator(ep); w.train_discriminator(ep)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    w.train_generator(17); w.train_discriminator(17)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::detach", "aten::alias",
                                                        "aten::reshape", "aten::_unsafe_view", "aten::unsqueeze", "aten::squeeze", "aten::select", "aten::slice", "aten::transpose", "aten::t",
                                                        "aten::permute", "aten::expand", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::lift_fresh",
                                                        "aten::resolve_conj", "aten::resolve_neg", "aten::unbind", "aten::stride", "aten::size", "aten::view_as", "aten::flatten", "aten::narrow", "aten::unflatten"):
        site, par = "?", ev.cpu_parent
        while par is not None:                       # backward ops run on the autograd thread: name the node being evaluated
            if "evaluate_function" in par.name or par.name.endswith("Backward"):
                site = par.name.replace("autograd::engine::evaluate_function: ", "")[:48]
                break
            par = par.cpu_parent
        for fr in ev.stack:
            if "/lcgan_amd/" in fr:
                site = fr.split("/lcgan_amd/")[-1].split("(")[0].strip()[:60]
                break
        shape = str(ev.input_shapes[0])[:28] if ev.input_shapes else ""
        cnt[(ev.name, site, shape)] += 1
for k, v in cnt.most_common(120):
    print(v, k)
print("---- large operands (>= 1M elements at this batch)")
big = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::mul") and ev.input_shapes and ev.input_shapes[0]:
        n = 1
        for d in ev.input_shapes[0]:
            n *= d
        if n >= (1 << 20):
            par, chain = ev.cpu_parent, []
            while par is not None and len(chain) < 4:
                chain.append(par.name.replace("autograd::engine::evaluate_function: ", "")[:40])
                par = par.cpu_parent
            big[(ev.name, str(ev.input_shapes[0]), " < ".join(chain))] += 1
for k, v in big.most_common(40):
    print(v, k)
