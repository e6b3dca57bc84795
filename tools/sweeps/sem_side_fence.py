"""Does an event record + wait on the side stream behind every module of the semantic head (barrier packets with full
fences between its kernels) make the side-stream head reproducible?  (tools/sweeps/stream_soak.py with hooks)"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "fence"
mcnn.SEM_SIDE_STREAM = True
if mode == "fence":
    def fence(mod, inp, out):
        st = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(st)
        st.wait_event(ev)
    for m in model.sem_seg_head.modules():
        if not list(m.children()):
            m.register_forward_hook(fence)
ref, bad = None, 0
for it in range(20):
    losses = model(inputs)          # forward only: the anomaly shows in loss_sem_seg
    v = float(losses["loss_sem_seg"])
    torch.cuda.synchronize()
    if ref is None:
        ref = v
    elif v != ref:
        bad += 1
print("mode %s: %d of 19 forwards differ from the first (loss_sem_seg %.7f)" % (mode, bad, ref))
