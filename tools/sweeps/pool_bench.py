"""Time the multi-level MOIPool / ROIAlign forward and backward on the bench's roi sets (clustered and uniform)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.structures import Boxes
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
CL = torch.channels_last


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for cluster in (1.0, 0.0):
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=cluster, objects=40)
    feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL).requires_grad_() for s in (4, 8, 16, 32)]
    boxes = [x["proposals"].proposal_boxes for x in inputs]
    oh = [x["proposals"].oh_labels for x in inputs]
    sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
    moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
    out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
    g = torch.randn_like(out)
    t_f = timeit(lambda: moi(feats, boxes, oh_labels_list=oh, superpixels=sp))
    t_fb = timeit(lambda: torch.autograd.grad(moi(feats, boxes, oh_labels_list=oh, superpixels=sp)[0], feats, g))
    print("cluster %.1f  MOIPool fwd %.0f us  bwd %.0f us  (valid argmax %.3f)" % (cluster, t_f, t_fb - t_f, float((arg >= 0).float().mean())))
    for nfg in (300, 60):
        sel = [Boxes(b.tensor[:nfg // 2]) for b in boxes]
        al = ROIPooler(14, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "ROIAlignV2")
        o = al(feats, sel)
        g2 = torch.randn_like(o)
        t_f = timeit(lambda: al(feats, sel))
        t_fb = timeit(lambda: torch.autograd.grad(al(feats, sel), feats, g2))
        print("cluster %.1f  ROIAlign 14x14 on %d rois: fwd %.0f us  bwd %.0f us" % (cluster, nfg, t_f, t_fb - t_f))

# fixed cost of the backward gathers: a handful of rois on the same maps
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=0.0, objects=40)
feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL).requires_grad_() for s in (4, 8, 16, 32)]
sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
for nr in (4, 64, 500):
    boxes = [Boxes(x["proposals"].proposal_boxes.tensor[:nr]) for x in inputs]
    oh = [x["proposals"].oh_labels[:nr] for x in inputs]
    moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
    out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
    g = torch.randn_like(out)
    t_f = timeit(lambda: moi(feats, boxes, oh_labels_list=oh, superpixels=sp))
    t_fb = timeit(lambda: torch.autograd.grad(moi(feats, boxes, oh_labels_list=oh, superpixels=sp)[0], feats, g))
    print("MOIPool with %d rois per image: fwd %.0f us  bwd %.0f us" % (nr, t_f, t_fb - t_f))
