"""Soak test of the side streams: the same step (no optimizer) N times, every loss and every gradient compared bit for
bit with the first run.  usage: stream_soak.py [N]   (JTSM_WGRAD_STREAM / JTSM_SEM_SIDE_STREAM / JTSM_MOI_BWD_STREAMS in the env)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model_util import jtsm_cfg
from jtsm_amd.modeling import build_model
from jtsm_amd.utils.synthetic import synthetic_inputs
cuda = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(jtsm_cfg("cuda")); model.train(); model.roi_heads.box_head.dropout_p = 0.0
with torch.no_grad():
    model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda, cluster=1.0, objects=40)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
from jtsm_amd.layers import conv as K
ref, bad_runs = None, 0
side_default = K.WGRAD_STREAM
for it in range(N):
    K.WGRAD_STREAM = side_default and it > 0          # the first run (the reference) on ONE stream
    model.zero_grad(set_to_none=True)
    losses = model(inputs)
    sum(losses.values()).backward()
    cur = {"loss/" + k: v.detach().clone() for k, v in losses.items()}
    cur.update({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    bad = [k for k in ref if not torch.equal(ref[k], cur[k])]
    if bad:
        bad_runs += 1
        print("run %d: %d tensors differ, e.g. %s" % (it, len(bad), bad[:4]), flush=True)
print("soak: %d of %d runs differ from the first" % (bad_runs, N - 1))
