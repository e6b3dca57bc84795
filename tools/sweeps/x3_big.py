"""Forward / data-gradient timings of the layers that run the 256 x 256-tile bf16x3 kernels (diagnostic builds)."""
import sys, torch
sys.path.insert(0, '.')
from jtsm_amd.layers import conv as K
CL = torch.channels_last
cuda = torch.device('cuda:0')
def kernel_ms(f, n=10):
    K.LAUNCH_LOG = []
    for _ in range(n): f()
    torch.cuda.synchronize()
    t = sorted(sp.kernel_ms() for (_, _, sp, _, _) in K.LAUNCH_LOG)
    K.LAUNCH_LOG = None
    return t[len(t) // 2]
shapes = [((255, 256, 14, 14, 256, 3, 1, 1)), ((4000, 12544, 1, 1, 2048, 1, 1, 0)), ((4000, 2048, 1, 1, 4096, 1, 1, 0))]
for shp in shapes:
    (N, C, H, W, O, k, s, p) = shp
    x = torch.randn(N, C, H, W, device=cuda).contiguous(memory_format=CL)
    w = (torch.randn(O, C, k, k, device=cuda) * 0.05).contiguous(memory_format=CL)
    y = K.conv2d_forward(x, w, s, p, 1)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * C * k * k
    a = kernel_ms(lambda: K.conv2d_forward(x, w, s, p, 1))
    b = kernel_ms(lambda: K.conv2d_backward_data(dy, w, tuple(x.shape), s, p, 1))
    print("%-40s fwd %.3f (%.0f TF) dgrad %.3f (%.0f TF)" % (str(shp), a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
