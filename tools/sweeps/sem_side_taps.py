"""The wrong elements of the side-stream anomaly as sums of the up-sampling's four terms: which subset of
w_k * tap_k (k = 00, 01, 10, 11) reproduces each wrong value?  Forward only, no kernels added to the side stream.
usage: sem_side_taps.py [N]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
outs = []
for name, m in model.sem_seg_head.named_modules():
    if name:
        m.register_forward_hook(lambda mod, inp, out, name=name: outs.append((name, out.detach())) if isinstance(out, torch.Tensor) else None)


def taps(x, n, c, h, w):
    H, W = x.shape[2], x.shape[3]
    def lerp(o, size):
        s = ((o.float() + 0.5) * 0.5 - 0.5).clamp(min=0)
        i0 = s.floor().long().clamp(max=size - 1)
        i1 = (i0 + 1).clamp(max=size - 1)
        return i0, i1, s - i0.float()
    h0, h1, lh = lerp(h, H)
    w0, w1, lw = lerp(w, W)
    return ([x[n, c, h0, w0], x[n, c, h0, w1], x[n, c, h1, w0], x[n, c, h1, w1]],
            [(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw])


ref = None
for it in range(N + 1):
    mcnn.SEM_SIDE_STREAM = it > 0
    del outs[:]
    losses = model(inputs)
    torch.cuda.synchronize()
    cur = [(n, t.clone()) for n, t in outs]
    if ref is None:
        ref = cur
        continue
    names = [n for n, _ in cur]
    bad = [k for k, ((n, a), (_, b)) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
    if not bad:
        print("run %d: equal" % it, flush=True)
        continue
    k = bad[0]
    name, a = cur[k]
    x = cur[k - 1][1]                       # the module in front (equal to the reference run's: it is not in `bad`)
    b = ref[k][1]
    wrong = (a != b)
    idx = wrong.nonzero()
    n, c, h, w = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    got, want = a[wrong], b[wrong]
    line = "run %d: %s behind %s, %d wrong, channel mod 4 %s;" % (it, name, names[k - 1], int(wrong.sum()), [int((c % 4 == j).sum()) for j in range(4)])
    if x.shape[2] * 2 != a.shape[2]:
        print(line, "not an up-sampling", flush=True)
        continue
    v, wt = taps(x, n, c, h, w)
    full = sum(p * q for p, q in zip(v, wt))
    line += " tap model on the right values: max err %.1e;" % float((full - want).abs().max())
    expl = torch.zeros_like(got, dtype=torch.bool)
    for mask in range(15):
        part = sum(v[j] * wt[j] for j in range(4) if mask >> j & 1) if mask else torch.zeros_like(got)
        hit = (got - part).abs() <= 2e-6 * (1 + part.abs())
        if int(hit.sum()) > len(got) // 50:
            line += " terms %s: %d;" % ([j for j in range(4) if mask >> j & 1], int(hit.sum()))
        expl |= hit
    line += " explained by some subset of terms: %d of %d" % (int(expl.sum()), len(got))
    print(line, flush=True)
    piece = ((n * a.shape[2] + h) * a.shape[3] + w) * (a.shape[1] // 4) + c // 4
    lane, wave = piece % 64, piece // 64
    lh = torch.bincount((lane // 8).long(), minlength=8).tolist()
    uw, cnt = torch.unique(wave, return_counts=True)
    blocks = torch.unique(piece // 256)
    print("     lanes (groups of 8): %s | distinct waves %d (of %d), wrong elements per such wave: min %d median %d max %d | distinct workgroups %d; first wrong workgroups %s | h parity %s w parity %s" % (
        lh, len(uw), a.numel() // 256, int(cnt.min()), int(cnt.median()), int(cnt.max()), len(blocks), blocks[:8].tolist(),
        torch.bincount((h % 2).long(), minlength=2).tolist(), torch.bincount((w % 2).long(), minlength=2).tolist()), flush=True)
    w0 = uw[0]
    sel = wave == w0
    print("     first such wave %d: lanes %s components %s" % (int(w0), lane[sel].tolist()[:40], (c[sel] % 4).tolist()[:40]), flush=True)
    from jtsm_amd import _lib as L
    import ctypes as C
    if hasattr(L.lib(), "jtsm_diag_up2_read"):
        buf = (C.c_uint * 8)()
        L.lib().jtsm_diag_up2_read(buf, 8)
        print("     self-check counters since the last read: pieces %d, operands differ %d, result differs %d, booked %d" % (buf[0], buf[1], buf[2], buf[4]), flush=True)
