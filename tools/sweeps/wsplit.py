"""One weight-gradient shape (argv[1] = index) timed with the shipped split-K plan: kernel and whole call.

Run from the repo root on the GPU box:  python tools/sweeps/wsplit.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, os, torch
sys.path.insert(0, '.')
from jtsm_amd.layers import conv as K
CL = torch.channels_last
cuda = torch.device('cuda:0')
def times(f, n=10):
    K.LAUNCH_LOG = []
    for _ in range(n): f()
    torch.cuda.synchronize()
    k = sorted(sp.kernel_ms() for (_, _, sp, _) in K.LAUNCH_LOG); c = sorted(sp.call_ms() for (_, _, sp, _) in K.LAUNCH_LOG)
    K.LAUNCH_LOG = None
    return k[len(k)//2], c[len(c)//2]
shapes = [(2, 256, 64, 64, 1024, 1, 1, 0), (2, 1024, 64, 64, 256, 1, 1, 0), (2, 128, 128, 128, 512, 1, 1, 0), (12, 256, 14, 14, 256, 3, 1, 1), (2, 512, 32, 32, 2048, 1, 1, 0)]
N, C, H, W, O, k, s, p = shapes[int(sys.argv[1])]
x = torch.randn(N, C, H, W, device=cuda).contiguous(memory_format=CL)
w = (torch.randn(O, C, k, k, device=cuda) * 0.05).contiguous(memory_format=CL)
Ho = (H + 2 * p - k) // s + 1
dy = torch.randn(N, O, Ho, Ho, device=cuda).contiguous(memory_format=CL)
fl = 2.0 * dy.numel() * C * k * k
K._PLANS.clear()
tk, tc = times(lambda: K.conv2d_backward_weight(dy, x, tuple(w.shape), s, p, 1))
print("%s split=%s kernel %.3f ms call %.3f ms  %.0f TF(call)" % (str(shapes[int(sys.argv[1])]), os.environ.get("JTSM_WSPLIT", "auto"), tk, tc, fl / tc / 1e9), flush=True)
