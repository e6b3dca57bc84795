"""MOIPool + ROIAlign forward / backward on the bench's clustered rois, three times (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.structures import Boxes
from jtsm_amd.utils.synthetic import synthetic_inputs
dev = torch.device("cuda", 0)
CL = torch.channels_last
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL).requires_grad_() for s in (4, 8, 16, 32)]
boxes = [x["proposals"].proposal_boxes for x in inputs]
oh = [x["proposals"].oh_labels for x in inputs]
sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
al = ROIPooler(14, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "ROIAlignV2")
sel = [Boxes(b.tensor[:150]) for b in boxes]
for _ in range(3):
    out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
    torch.autograd.grad(out, feats, torch.ones_like(out))
    o = al(feats, sel)
    torch.autograd.grad(o, feats, torch.ones_like(o))
torch.cuda.synchronize()
