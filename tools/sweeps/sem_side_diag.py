"""Reads the self-check counters of the diagnostic up-sampling kernel (csrc/semseg_ops.hip under -DJTSM_DIAG_UP2,
library selected with JTSM_HIP_LIB) after forward passes with the semantic head on one stream and on its own.
usage: JTSM_HIP_LIB=scratch/alt/sem_diag.so python tools/sweeps/sem_side_diag.py [N]"""
import ctypes as C, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model_util import jtsm_cfg
from jtsm_amd import _lib as L
from jtsm_amd.modeling import build_model
from jtsm_amd.modeling.meta_arch import mcnn
from jtsm_amd.utils.synthetic import synthetic_inputs
cuda = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(jtsm_cfg("cuda")); model.train(); model.roi_heads.box_head.dropout_p = 0.0
with torch.no_grad():
    model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda, cluster=1.0, objects=40)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WORDS = 8 + 64 * 40
buf = (C.c_uint * WORDS)()
lib = L.lib()
outs = []
for name, m in model.sem_seg_head.named_modules():
    if name:
        m.register_forward_hook(lambda mod, inp, out, name=name: outs.append((name, out.detach())) if isinstance(out, torch.Tensor) else None)


def f(u):
    return struct.unpack("f", struct.pack("I", u))[0]


ref = None
for it in range(N + 1):
    mcnn.SEM_SIDE_STREAM = it > 0
    del outs[:]
    losses = model(inputs)
    torch.cuda.synchronize()
    assert lib.jtsm_diag_up2_read(buf, WORDS) == 0
    cur = [(n, t.clone()) for n, t in outs]
    if ref is None:
        ref = cur
    bad = [(n, int((a != b).sum())) for (n, a), (_, b) in zip(cur, ref) if not torch.equal(a, b)]
    print("run %d side %d: pieces %d, operands differ on second look %d, result differs %d (of them with equal operands %d), booked %d, stored piece differs on read-back %d, packed result differs from a second evaluation BEFORE the store %d | tensors differing from run 0: %s" % (
        it, it > 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], bad[:3]), flush=True)
    for s in range(min(buf[4], 6 if it < 3 else 2)):
        d = buf[8 + s * 40: 8 + (s + 1) * 40]
        v, r, o = [f(x) for x in d[4:20]], [f(x) for x in d[20:36]], [f(x) for x in d[36:40]]
        diff = [k for k in range(16) if d[4 + k] != d[20 + k]]
        print("   piece %d lane %d block %d flags %d: operand slots that differ %s first %s second %s | out %s" % (
            d[0], d[1] % 64, d[2], d[3], diff, ["%.4f" % v[k] for k in diff], ["%.4f" % r[k] for k in diff], ["%.4f" % x for x in o]), flush=True)
