import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from model_util import jtsm_cfg
from jtsm_amd.layers import conv as K, fused_blocks
from jtsm_amd.modeling import build_model
from jtsm_amd.utils.synthetic import synthetic_inputs
calls = []
orig = fused_blocks.identity_chain_fused
def spy(x, blocks):
    calls.append((tuple(x.shape), len(blocks)))
    return orig(x, blocks)
fused_blocks.identity_chain_fused = spy
K.set_math("f16")
torch.manual_seed(0)
m = build_model(jtsm_cfg("cuda", depth=101)); m.train()
with torch.no_grad():
    m.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
inputs = synthetic_inputs(1234, batch=2, size=1024, width=2048, proposals=2000, device=torch.device("cuda"), cluster=1.0, objects=40)
for chain in (True, False, True, False):
    fused_blocks.CHAIN16 = chain
    del calls[:]
    for _ in range(2):
        m.zero_grad(set_to_none=True); sum(m(inputs).values()).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.zero_grad(set_to_none=True); sum(m(inputs).values()).backward()
    torch.cuda.synchronize()
    print("chain", chain, "%.2f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3), "calls per step:", calls[:4])
