"""MOIPool forward A/B on the bench's roi sets: one process per kernel form (the library reads JTSM_MOI_FWD_ROWS once),
hipEvent timing of the forward call (bits + pooling kernels) and an exact checksum of values and arg-max:
    for m in 0 1 2 3 4; do JTSM_MOI_FWD_ROWS=$m python tools/sweeps/moi_fwd_ab.py; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
CL = torch.channels_last
mode = os.environ.get("JTSM_MOI_FWD_ROWS", "1")
torch.manual_seed(0)
for cluster in (1.0, 0.0):
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=cluster, objects=40)
    feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL) for s in (4, 8, 16, 32)]
    boxes = [x["proposals"].proposal_boxes for x in inputs]
    oh = [x["proposals"].oh_labels for x in inputs]
    sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
    if os.environ.get("SORT", "0") != "0":     # rois of an image in spatial order: level, then coarse centre (row-major)
        from jtsm_amd.structures import Boxes
        g = int(os.environ.get("SORT"))
        for i in range(len(boxes)):
            t = boxes[i].tensor
            lvl = torch.floor(4 + torch.log2(torch.sqrt((t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])) / 224 + 1e-8)).clamp(2, 5)
            cx, cy = ((t[:, 0] + t[:, 2]) / 2 / g).floor(), ((t[:, 1] + t[:, 3]) / 2 / g).floor()
            key = (lvl * 64 + cy) * 64 + cx
            order = torch.argsort(key, stable=True)
            boxes[i] = Boxes(t[order].contiguous())
            oh[i] = oh[i][order].contiguous()
    moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
    with torch.no_grad():
        for _ in range(3):
            out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
        b.record()
        torch.cuda.synchronize()
    chk = (int(out.view(torch.int32).to(torch.int64).sum()), int(arg.to(torch.int64).sum()), int((arg >= 0).sum()))
    print("sort %s mode %s cluster %.0f: %.1f us per forward call, checksum %s" % (os.environ.get("SORT", "0"), mode, cluster, a.elapsed_time(b) * 50, chk), flush=True)
