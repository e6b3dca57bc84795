"""Times the fused x4-upsample + cross-entropy backward at the JTSM size (2 x 54 x 256 x 256 logits), both forms."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jtsm_amd.layers.elementwise import semseg_cross_entropy  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
z = (torch.randn(2, 56, 256, 256, generator=g) * 3).to(dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 54, (2, 1024, 1024), generator=g).to(dev)
for form in ("0", "1"):
    os.environ["JTSM_CE_BWD_TILED"] = form
    zd = z.clone().requires_grad_()
    for rep in range(3):
        loss = semseg_cross_entropy(zd[:, :54], t, 4, 255)
        zd.grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            loss.backward(retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
    print("tiled=%s: %.1f us per backward (incl. autograd overhead)" % (form, e0.elapsed_time(e1) * 50), flush=True)
