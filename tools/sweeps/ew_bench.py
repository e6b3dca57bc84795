"""Micro-timings of bandwidth kernels on shapes of the bench step (channel_sum, ROIAlign backward on piled rois)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jtsm_amd.layers.elementwise import channel_sum
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.structures import Boxes

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for rows, c in [] if os.environ.get("EW_SKIP_CSUM") else [(321 * 784, 80), (321 * 784, 256), (321 * 196, 1024), (321 * 196, 256), (4000, 2048), (4000, 4096),
                (4000, 1872), (2 * 256 * 256, 256), (2 * 256 * 256, 128), (2 * 128 * 128, 256), (2 * 256 * 256, 56)]:
    g = torch.randn(rows, c, device=dev)
    ms = timeit(lambda: channel_sum(g))
    print("channel_sum rows=%7d C=%4d  %.3f ms  %.0f GB/s" % (rows, c, ms, rows * c * 4 / ms / 1e6))

feats = [torch.randn(2, 256, 256 // s, 256 // s, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for s in (1, 2, 4, 8)]
pool = ROIPooler(14, (0.25, 0.125, 0.0625, 0.03125), 0, "ROIAlignV2")
g = torch.Generator().manual_seed(0)
for piles, per in [(6, 54), (6, 10), (40, 8), (320, 1)]:
    boxes = []
    for i in range(2):
        bs = []
        for p in range(piles // 2):
            x0, y0 = float(torch.rand(1, generator=g) * 600), float(torch.rand(1, generator=g) * 600)
            w, h = float(torch.rand(1, generator=g) * 300 + 60), float(torch.rand(1, generator=g) * 300 + 60)
            j = (torch.rand(per, 4, generator=g) - 0.5) * 0.24
            b = torch.tensor([x0, y0, x0 + w, y0 + h]) + j * torch.tensor([w, h, w, h])
            bs.append(b)
        boxes.append(Boxes(torch.cat(bs).to(dev)))
    out = pool(feats, boxes)
    go = torch.randn_like(out)

    def bw():
        for f in feats:
            f.grad = None
        out.backward(go, retain_graph=True)
    print("roi_align 14x14 bwd: %d piles x %d rois  %.3f ms" % (piles, per, timeit(bw, 10)))
