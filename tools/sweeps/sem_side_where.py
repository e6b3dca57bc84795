"""Where does the semantic head's side-stream run first differ from the one-stream run?  Every module output of the
head is kept (no extra kernels on the stream: references only) and compared after the step; then the gradients.
usage: sem_side_where.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model_util import jtsm_cfg
from jtsm_amd.modeling import build_model
from jtsm_amd.modeling.meta_arch import mcnn
from jtsm_amd.utils.synthetic import synthetic_inputs
cuda = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(jtsm_cfg("cuda")); model.train(); model.roi_heads.box_head.dropout_p = 0.0
with torch.no_grad():
    model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda, cluster=1.0, objects=40)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BACKWARD = os.environ.get("WHERE_BACKWARD", "1") != "0"
outs = []
for name, m in model.sem_seg_head.named_modules():
    if name:
        m.register_forward_hook(lambda mod, inp, out, name=name: outs.append((name, out.detach())) if isinstance(out, torch.Tensor) else None)
ref = None
for it in range(N + 1):
    mcnn.SEM_SIDE_STREAM = it > 0
    del outs[:]
    model.zero_grad(set_to_none=True)
    losses = model(inputs)
    if BACKWARD:
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    cur = [(n, t.clone()) for n, t in outs]
    cur += [("loss/" + k, v.detach().clone()) for k, v in losses.items()]
    if BACKWARD:
        cur += [("grad/" + n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.grad is not None]
    if ref is None:
        ref = cur
        continue
    bad = [(n, int((a != b).sum()), a.numel()) for (n, a), (_, b) in zip(cur, ref) if not torch.equal(a, b)]
    print("run %d: %d tensors differ; first: %s" % (it, len(bad), bad[:6]), flush=True)
