"""Weight-gradient contraction timings over the layer shapes of the bench model (kernel-only, library event hook).

Run from the repo root on the GPU box:  python tools/sweeps/wgrad_sweep.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, torch
sys.path.insert(0, '.')
from jtsm_amd.layers import conv as K
CL = torch.channels_last
cuda = torch.device('cuda:0')
def kernel_ms(f, n=10):
    K.LAUNCH_LOG = []
    for _ in range(n): f()
    torch.cuda.synchronize()
    t = sorted(sp.kernel_ms() for (_, _, sp, _) in K.LAUNCH_LOG)
    K.LAUNCH_LOG = None
    return t[len(t) // 2]
shapes = [(2, 256, 64, 64, 256, 3, 1, 1), (2, 128, 128, 128, 128, 3, 1, 1), (2, 512, 32, 32, 512, 3, 1, 1), (2, 256, 64, 64, 1024, 1, 1, 0),
          (2, 1024, 64, 64, 256, 1, 1, 0), (2, 128, 128, 128, 512, 1, 1, 0), (10, 256, 14, 14, 256, 3, 1, 1), (2, 512, 32, 32, 2048, 1, 1, 0),
          (2, 256, 256, 256, 256, 3, 1, 1), (4000, 12544, 1, 1, 2048, 1, 1, 0), (4000, 4096, 1, 1, 1872, 1, 1, 0)]
tot = 0
for (N, C, H, W, O, k, s, p) in shapes:
    x = torch.randn(N, C, H, W, device=cuda).contiguous(memory_format=CL)
    w = (torch.randn(O, C, k, k, device=cuda) * 0.05).contiguous(memory_format=CL)
    Ho = (H + 2 * p - k) // s + 1
    dy = torch.randn(N, O, Ho, Ho if W > 1 else 1, device=cuda).contiguous(memory_format=CL)
    fl = 2.0 * dy.numel() * C * k * k
    t = kernel_ms(lambda: K.conv2d_backward_weight(dy, x, tuple(w.shape), s, p, 1))
    tot += t
    print("%-40s %.3f ms %.0f TF" % (str((N, C, H, W, O, k, s, p)), t, fl / t / 1e9), flush=True)
print("sum", tot)
