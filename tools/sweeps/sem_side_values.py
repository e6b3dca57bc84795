"""What ARE the wrong values of the side-stream anomaly?  For the first differing tensor of a side-stream forward:
which channels / positions, and whether each wrong value occurs (bit pattern) in the reference run's tensors of the
head — the same tensor elsewhere (misplaced), another tensor (old contents of the block: a lost write) or nowhere.
usage: sem_side_values.py [N]"""
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
ref = None
for it in range(N + 1):
    mcnn.SEM_SIDE_STREAM = it > 0
    del outs[:]
    losses = model(inputs)
    torch.cuda.synchronize()
    ptrs = {n: (t.data_ptr(), t.numel() * 4) for n, t in outs}
    cur = [(n, t.clone()) for n, t in outs]
    if ref is None:
        ref = cur
        ref_ptrs = ptrs
        continue
    bad = [(n, a, b) for (n, a), (_, b) in zip(cur, ref) if not torch.equal(a, b)]
    if not bad:
        print("run %d: equal" % it, flush=True)
        continue
    name, a, b = bad[0]
    a_m, b_m = a.permute(0, 2, 3, 1).contiguous(), b.permute(0, 2, 3, 1).contiguous()    # memory (NHWC) order
    wrong = (a_m != b_m)
    idx = wrong.nonzero()
    c = idx[:, 3]
    got, want = a_m[wrong], b_m[wrong]
    print("run %d: first differing %s (%d elements of %d) at %x; channel histogram mod 4: %s; channels < 64: %d; got range %.3f..%.3f" % (
        it, name, int(wrong.sum()), a.numel(), ptrs[name][0], [int((c % 4 == k).sum()) for k in range(4)], int((c < 64).sum()),
        float(got.min()), float(got.max())), flush=True)
    gi = got.view(torch.int32)
    for n2, t in ref:
        if t.dtype != torch.float32 or t.dim() != 4:
            continue
        hit = torch.isin(gi, t.view(torch.int32).flatten())
        nz = hit & (got != 0)
        print("     wrong values found in the reference run's %-10s: %5d (non-zero ones %5d)" % (n2, int(hit.sum()), int(nz.sum())), flush=True)
    # where did each tensor of this run live, relative to the reference run?  (block reuse)
    owner = [n2 for n2, (p, sz) in ref_ptrs.items() if p <= ptrs[name][0] < p + sz]
    print("     this tensor's address held in the reference run:", owner, "| zeros among wrong values:", int((got == 0).sum()), flush=True)
