"""Statistics of the semantic head's side-stream anomaly (DESIGN §5 dead end 38): N forward passes with the head on
its own stream; for the up-sampling behind p3's GroupNorm, the number of elements that differ from a torch
up-sampling of the SAME input (cloned in stream order), which channels they are, and what the wrong values are made of
(the four taps a, b, c, d of the element with weights wa..wd: which partial sums reproduce the wrong value).
usage: sem_side_stats.py [N]      (JTSM_HIP_LIB selects a diagnostic build of the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
keep = {}


def hook(mod, inp, out):
    keep["inp_copy"], keep["out_copy"] = inp[0].detach().clone(), out.detach().clone()


dict(model.sem_seg_head.named_modules())["p3.1"].register_forward_hook(hook)


def taps(x, n, c, h, w):
    """The four source values and weights of output element (n, c, h, w) of a 2x bilinear up-sampling (align_corners
    False) of x."""
    H, W = x.shape[2], x.shape[3]
    def lerp(o, size):
        s = ((o.double() + 0.5) * 0.5 - 0.5).clamp(min=0)
        i0 = s.floor().long().clamp(max=size - 1)
        i1 = (i0 + 1).clamp(max=size - 1)
        return i0, i1, (s - i0).float()
    h0, h1, lh = lerp(h, H)
    w0, w1, lw = lerp(w, W)
    vals = [x[n, c, h0, w0], x[n, c, h0, w1], x[n, c, h1, w0], x[n, c, h1, w1]]
    wts = [(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw]
    return vals, wts


bad_runs, total_wrong = 0, 0
for it in range(N + 1):
    mcnn.SEM_SIDE_STREAM = it > 0
    losses = model(inputs)
    torch.cuda.synchronize()
    x, o = keep["inp_copy"], keep["out_copy"]
    want = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    wrong = (o - want).abs() > 1e-4
    nw = int(wrong.sum())
    if it == 0:
        print("one stream: wrong elements", nw, flush=True)
        continue
    if nw == 0:
        continue
    bad_runs += 1
    total_wrong += nw
    idx = wrong.nonzero()
    n, c, h, w = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    got = o[wrong]
    v, wt = taps(x, n, c, h, w)
    full = sum(a * b for a, b in zip(v, wt))
    line = "run %d: %d wrong; odd channels %d; channels < 64: %d;" % (it, nw, int((c % 2 == 1).sum()), int((c < 64).sum()))
    # which subset of the four terms reproduces the wrong value?
    for mask in range(15):
        part = sum(v[k] * wt[k] for k in range(4) if mask >> k & 1) if mask else torch.zeros_like(got)
        hit = int(((got - part).abs() < 1e-5).sum())
        if hit > nw // 20:
            line += " terms%s: %d;" % ([k for k in range(4) if mask >> k & 1], hit)
    # a term taken from the neighbouring channel (the other half of a packed pair)?
    for dc in (-1, 1, 2, -2):
        cc = (c + dc).clamp(0, x.shape[1] - 1)
        v2, _ = taps(x, n, cc, h, w)
        for k in range(4):
            alt = full - v[k] * wt[k] + v2[k] * wt[k]
            hit = int(((got - alt).abs() < 1e-5).sum())
            if hit > nw // 20:
                line += " tap%d from channel%+d: %d;" % (k, dc, hit)
    print(line, "| check of the tap model on right elements: max err %.2e" % float((full - want[wrong]).abs().max()), flush=True)
print("stats: %d of %d side-stream runs wrong, %d elements in all" % (bad_runs, N, total_wrong))
