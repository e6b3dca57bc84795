"""Diagnostic: per-loss relative deviation of the DC5 composite from the oracle, per conv arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_hip_wsl_v2 import dc5_cfg
from model_util import to_batched_inputs
from oracle import model as OM
from jtsm_amd.layers import conv as K
from jtsm_amd.modeling import build_model
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dan = (4096, 4096) if depth == 18 else (2048, 4096)
params = OM.init_params_dc5(seed=11, depth=depth, nt=20, ns=2, dan_dims=dan, input_gain=1.0 / 64)
batch = OM.synthetic_batch(77, B=2, size=256, R=120, sp_block=8, n_stuff=1, nt=20, ns=2)
losses0, aux0 = OM.forward_losses(params, batch, depth=depth, return_aux=True, arch="dc5", nt=20, ns=2)
for math in ("f32", "bf16x3", "f32", "bf16x3"):
    K.set_math(math)
    model = build_model(dc5_cfg("cuda", depth))
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.train(); model.roi_heads.box_head.dropout_p = 0.0
    losses = model(to_batched_inputs(batch))
    dev = {k: abs(float(losses[k].detach()) - float(losses0[k])) / max(abs(float(losses0[k])), 1e-6) for k in losses0}
    a = model.roi_heads.aux["pooled_argmax"].cpu().contiguous(); b = aux0["pooled_argmax"]
    print(math, "worst", max(dev.values()), sorted(dev.items(), key=lambda kv: -kv[1])[:4], "argmax flips", (a != b).float().mean().item(), flush=True)
