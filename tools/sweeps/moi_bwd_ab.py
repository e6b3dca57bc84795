"""MOIPool backward A/B on the bench's roi sets (hipEvent timing of the backward call; set the JTSM_MOI_* switches in
the environment, one process per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
CL = torch.channels_last
torch.manual_seed(0)
for cluster in (1.0, 0.0):
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=cluster, objects=40)
    feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev).contiguous(memory_format=CL).requires_grad_() for s in (4, 8, 16, 32)]
    boxes = [x["proposals"].proposal_boxes for x in inputs]
    oh = [x["proposals"].oh_labels for x in inputs]
    sp = torch.stack([x["superpixels"] for x in inputs]).to(dev)
    moi = ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32), 0, "MOIPool")
    out, arg = moi(feats, boxes, oh_labels_list=oh, superpixels=sp)
    g = torch.randn_like(out)
    for _ in range(3):
        grads = torch.autograd.grad(out, feats, g, retain_graph=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        grads = torch.autograd.grad(out, feats, g, retain_graph=True)
    b.record()
    torch.cuda.synchronize()
    chk = [int(x.view(torch.int32).to(torch.int64).sum()) for x in grads]
    print("cluster %.0f: %.1f us per backward call, checksum %s" % (cluster, a.elapsed_time(b) * 50, chk), flush=True)
