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
for mode, bwd in ((0, 0), (0, 0), (1, 0), (1, 0), (1, 1), (1, 1), (0, 1), (0, 1), (1, 1)):
    mcnn.SEM_SIDE_STREAM = bool(mode)
    model.zero_grad(set_to_none=True)
    losses = model(inputs)
    if bwd:
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    g = model.sem_seg_head.predictor.weight.grad
    print("side", mode, "bwd", bwd, "loss_sem_seg %.7f loss_mask %.7f loss_cls %.7f" % (float(losses["loss_sem_seg"]), float(losses["loss_mask"]), float(losses["loss_cls"])),
          "pred.w.grad %.6e" % (float(g.abs().sum()) if g is not None else 0.0), flush=True)
