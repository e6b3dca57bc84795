"""MOIPool multi-level backward, scatter vs gather form, as the rois concentrate on one spot (the census fallback).

Run from the repo root on the GPU box:  python tools/sweeps/moi_cluster.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, torch, time, ctypes as C
sys.path.insert(0, '.')
from jtsm_amd import _lib as L
dev = torch.device('cuda:0')
B, Cc, res = 2, 256, 7
shapes = [(B, Cc, 256 >> i, 256 >> i) for i in range(4)]
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
def run(rois, level, tiled):
    M = rois.shape[0]
    g = torch.randn(M, Cc, res, res, device=dev).contiguous(memory_format=torch.channels_last)
    # argmax: a valid cell inside each roi's box at its level (random), -1 nowhere
    lv = level.to(torch.int32)
    arg = torch.empty(M, Cc, res, res, dtype=torch.int32, device=dev).contiguous(memory_format=torch.channels_last)
    for l in range(4):
        sel = (lv == l).nonzero()[:, 0]
        if sel.numel() == 0: continue
        W = shapes[l][3]; s = scales[l]
        x0 = (rois[sel, 1] * s).round().clamp(0, W - 1); x1 = (rois[sel, 3] * s).round().clamp(0, W - 1)
        y0 = (rois[sel, 2] * s).round().clamp(0, W - 1); y1 = (rois[sel, 4] * s).round().clamp(0, W - 1)
        # bin (ph,pw) picks a cell inside its own bin range roughly: centre of the bin
        ph = torch.arange(res, device=dev).float()
        cy = (y0[:, None] + (ph[None] + 0.5) * ((y1 - y0 + 1) / res)[:, None]).floor().clamp(0, W - 1)
        cx = (x0[:, None] + (ph[None] + 0.5) * ((x1 - x0 + 1) / res)[:, None]).floor().clamp(0, W - 1)
        cell = (cy[:, :, None] * W + cx[:, None, :]).to(torch.int32)          # (n,7,7)
        arg[sel] = cell[:, None].expand(-1, Cc, -1, -1).contiguous(memory_format=torch.channels_last)
    grads = [torch.empty(sh, device=dev).contiguous(memory_format=torch.channels_last) for sh in shapes]
    nl = 4
    Hs = (C.c_int * nl)(*[sh[2] for sh in shapes]); Ws = (C.c_int * nl)(*[sh[3] for sh in shapes])
    ptrs = (C.c_void_p * nl)(*[t.data_ptr() for t in grads])
    sc = (C.c_float * nl)(*scales) if tiled else None
    lib = L.lib()
    ws = torch.empty(max(lib.jtsm_moi_pool_backward_levels_workspace_bytes(Hs, Ws, nl, B, M), 16), dtype=torch.uint8, device=dev)
    f = lambda: L.check(lib.jtsm_moi_pool_backward_levels_f32(L.ptr(g), L.ptr(rois), L.ptr(lv), L.ptr(arg), ptrs, Hs, Ws, sc, nl, B, Cc, M, res, res, 0,
                        L.ptr(ws) if tiled else None, C.c_size_t(ws.numel() if tiled else 0), L.stream()))
    for _ in range(3): f()
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2], ts[-1]
from jtsm_amd.modeling.poolers import assign_boxes_to_levels
from jtsm_amd.structures import Boxes
gen = torch.Generator().manual_seed(0)
def make(M, spread):
    size = torch.exp(torch.rand(M, generator=gen) * (6.2 - 3.4) + 3.4)          # 30 .. 490 px
    cx = 512 + (torch.rand(M, generator=gen) - 0.5) * spread; cy = 512 + (torch.rand(M, generator=gen) - 0.5) * spread
    b = torch.stack([cx - size / 2, cy - size / 2, cx + size / 2, cy + size / 2], 1).clamp(0, 1023)
    img = torch.randint(0, B, (M,), generator=gen).float()
    return torch.cat([img[:, None], b], 1).to(dev), b.to(dev)
for M, spread in [(4000, 1000), (4000, 300), (4000, 60), (4000, 0)]:
    rois, boxes = make(M, spread)
    lv = assign_boxes_to_levels([Boxes(boxes)], 2, 5, 224, 4)
    a, b2 = run(rois, lv, False), run(rois, lv, True)
    print("M=%d centres within %4d px: scatter median %.0f max %.0f us, gather median %.0f max %.0f us" % (M, spread, a[0], a[1], b2[0], b2[1]))
print("---- census max per case (fresh generator)")
gen = torch.Generator().manual_seed(0)
for M, spread in [(4000, 1000), (4000, 300), (4000, 60), (4000, 0)]:
    rois, boxes = make(M, spread)
    lv = assign_boxes_to_levels([Boxes(boxes)], 2, 5, 224, 4).to(torch.int32)
    nl = 4
    Hs = (C.c_int * nl)(*[sh[2] for sh in shapes]); Ws = (C.c_int * nl)(*[sh[3] for sh in shapes])
    lib = L.lib()
    nb = lib.jtsm_moi_pool_backward_levels_workspace_bytes(Hs, Ws, nl, B, M)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    g = torch.zeros(M, Cc, res, res, device=dev).contiguous(memory_format=torch.channels_last)
    arg = torch.full((M, Cc, res, res), -1, dtype=torch.int32, device=dev).contiguous(memory_format=torch.channels_last)
    grads = [torch.empty(sh, device=dev).contiguous(memory_format=torch.channels_last) for sh in shapes]
    ptrs = (C.c_void_p * nl)(*[t.data_ptr() for t in grads]); sc = (C.c_float * nl)(*scales)
    L.check(lib.jtsm_moi_pool_backward_levels_f32(L.ptr(g), L.ptr(rois), L.ptr(lv), L.ptr(arg), ptrs, Hs, Ws, sc, nl, B, Cc, M, res, res, 0, L.ptr(ws), C.c_size_t(nb), L.stream()))
    torch.cuda.synchronize()
    ints = ws.view(torch.int32)
    ntile = sum(B * ((sh[2] + 7) // 8) * ((sh[3] + 7) // 8) for sh in shapes)
    base = nl * B * (M + 1)
    cen = ints[base:base + ntile]
    print(spread, "census max", int(cen.max()), "stored", int(ints[base + ntile]), "levels", torch.bincount(lv.long(), minlength=4).tolist())
