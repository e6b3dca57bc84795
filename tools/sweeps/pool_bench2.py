"""Kernel-level timing of the multi-level pooling operators on the bench's roi sets (run under rocprofv3
--kernel-trace; tools/sweeps/pool_seq.py prints the per-call durations):  python tools/sweeps/pool_bench2.py [moi|align|all]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.structures import Boxes
from jtsm_amd.utils.synthetic import synthetic_inputs

what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
CL = torch.channels_last
for cluster in (1.0, 0.0):
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=cluster, objects=40)
    feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL).requires_grad_() for s in (4, 8, 16, 32)]
    boxes = [x["proposals"].proposal_boxes for x in inputs]
    oh = [x["proposals"].oh_labels for x in inputs]
    sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
    if what in ("moi", "all"):
        moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
        out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
        g = torch.randn_like(out)
        for _ in range(5):
            torch.autograd.grad(moi(feats, boxes, oh_labels_list=oh, superpixels=sp)[0], feats, g)
    if what in ("align", "all"):
        for nfg in (300, 60):
            sel = [Boxes(b.tensor[:nfg // 2]) for b in boxes]
            al = ROIPooler(14, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "ROIAlignV2")
            o = al(feats, sel)
            g2 = torch.randn_like(o)
            for _ in range(5):
                torch.autograd.grad(al(feats, sel), feats, g2)
    torch.cuda.synchronize()
