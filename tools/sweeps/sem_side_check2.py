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
sums = []
keep = {}
def hook(name):
    def f(mod, inp, out):
        if isinstance(out, torch.Tensor):
            sums.append((name, inp[0].detach().double().sum() if isinstance(inp[0], torch.Tensor) else None, out.detach().double().sum()))
            if name == "p3.1":
                keep["inp"], keep["inp_copy"] = inp[0].detach(), inp[0].detach().clone()
                keep["out"], keep["out_copy"] = out.detach(), out.detach().clone()
    return f
for n, m in model.sem_seg_head.named_modules():
    if n and "." not in n or n.count(".") == 1:
        m.register_forward_hook(hook(n))
ref = None
for mode in (0, 1, 1, 1, 1):
    mcnn.SEM_SIDE_STREAM = bool(mode)
    del sums[:]
    losses = model(inputs)
    torch.cuda.synchronize()
    vals = [(n, float(a) if a is not None else None, float(b)) for n, a, b in sums]
    if ref is None:
        ref = vals
    bad = [(n, a, b, ra, rb) for (n, a, b), (_, ra, rb) in zip(vals, ref) if a != ra or b != rb]
    import torch.nn.functional as F
    want = F.interpolate(keep["inp_copy"], scale_factor=2.0, mode="bilinear", align_corners=False)
    print("side", mode, "loss_sem_seg %.7f" % float(losses["loss_sem_seg"]), "first differing:", [b[0] for b in bad[:3]],
          "| out changed after the op:", bool((keep["out"] != keep["out_copy"]).any()), "inp changed:", bool((keep["inp"] != keep["inp_copy"]).any()),
          "| out_copy vs torch upsample of inp_copy: max err %.3e, wrong elements %d" % (float((keep["out_copy"] - want).abs().max()), int(((keep["out_copy"] - want).abs() > 1e-4).sum())),
          "ptr %x size %d" % (keep["out"].data_ptr(), keep["out"].numel() * 4), flush=True)
