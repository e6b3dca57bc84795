"""Forward / data-gradient bf16x3 contraction timings over the layer shapes of the bench model.

Run from the repo root on the GPU box:  python tools/sweeps/x3_sweep.py
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
# (shape, count per step fwd, count dgrad) roughly as in the model
shapes = [
  ((2, 64, 256, 256, 64, 3, 1, 1), 3, 0), ((2, 64, 256, 256, 256, 1, 1, 0), 4, 0), ((2, 256, 256, 256, 64, 1, 1, 0), 2, 0),
  ((2, 128, 128, 128, 128, 3, 1, 1), 6, 6), ((2, 128, 128, 128, 512, 1, 1, 0), 4, 4), ((2, 512, 128, 128, 128, 1, 1, 0), 3, 3),
  ((2, 256, 64, 64, 256, 3, 1, 1), 7, 7), ((2, 256, 64, 64, 1024, 1, 1, 0), 6, 6), ((2, 1024, 64, 64, 256, 1, 1, 0), 6, 6),
  ((2, 512, 32, 32, 512, 3, 1, 1), 3, 3), ((2, 512, 32, 32, 2048, 1, 1, 0), 3, 3), ((2, 2048, 32, 32, 512, 1, 1, 0), 2, 2),
  ((10, 256, 14, 14, 256, 3, 1, 1), 8, 8),
  ((2, 256, 256, 256, 256, 3, 1, 1), 1, 1), ((2, 256, 128, 128, 256, 3, 1, 1), 1, 1), ((2, 256, 256, 256, 128, 3, 1, 1), 1, 1),
  ((4000, 12544, 1, 1, 2048, 1, 1, 0), 1, 1), ((4000, 2048, 1, 1, 4096, 1, 1, 0), 1, 1), ((4000, 4096, 1, 1, 1872, 1, 1, 0), 1, 1),
]
tf = td = 0
for (shp, nf, nd) in shapes:
    (N, C, H, W, O, k, s, p) = shp
    x = torch.randn(N, C, H, W, device=cuda).contiguous(memory_format=CL)
    w = (torch.randn(O, C, k, k, device=cuda) * 0.05).contiguous(memory_format=CL)
    y = K.conv2d_forward(x, w, s, p, 1)
    dy = torch.randn_like(y)
    sc, bi = torch.rand(O, device=cuda) + 0.5, torch.randn(O, device=cuda)
    res = torch.randn_like(y)
    fl = 2.0 * y.numel() * C * k * k
    a = kernel_ms(lambda: K.conv2d_forward(x, w, s, p, 1, sc, bi, res, True))
    b = kernel_ms(lambda: K.conv2d_backward_data(dy, w, tuple(x.shape), s, p, 1))
    tf += a * nf; td += b * nd
    print("%-40s fwd %.3f (%.0f TF) dgrad %.3f (%.0f TF)" % (str(shp), a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
print("weighted sum fwd %.3f dgrad %.3f" % (tf, td))
