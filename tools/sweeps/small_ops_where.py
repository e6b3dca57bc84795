"""Which Python lines issue the small torch ops of a training step's FORWARD and optimizer (dispatch mode + traceback)."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
model = bench.build(dev)
opt = bench.make_optimizer(model)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
agg = collections.Counter()
elems = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
        if t is not None and t.is_cuda and not any(x in name for x in ("view", "reshape", "as_strided", "permute", "slice", "select", "detach", "alias", "expand", "unsqueeze", "squeeze", "transpose", "t.default", "split", "unbind", "empty", "_unsafe_view", "lift_fresh", "sym_", "is_", "stride", "size", "numel", "dim", "_local_scalar", "item", "unfold", "narrow", "chunk", "view_as", "_to_copy" if False else "@@")):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "jtsm_amd/" in fr.filename or fr.filename.endswith("bench.py") or "small_ops_where" in fr.filename:
                    where = "%s:%d" % (fr.filename[fr.filename.find("jtsm_amd/"):] if "jtsm_amd/" in fr.filename else os.path.basename(fr.filename), fr.lineno)
                    break
            agg[(name, where)] += 1
            elems[(name, where)] += t.numel()
        return out


def step(spy):
    if spy:
        with Spy():
            losses = model(inputs)
            total = sum(losses.values())
    else:
        losses = model(inputs)
        total = sum(losses.values())
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(2):
    step(False)
step(True)
print("forward device ops by call site (n, elements):")
for (name, where), n in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0])):
    print("n=%3d  %12d el  %-34s %s" % (n, elems[(name, where)], name, where))
