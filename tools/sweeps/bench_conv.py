"""fp32-MFMA contraction timings (exact-fp32 mode) over the bench shapes.

Run from the repo root on the GPU box:  python tools/sweeps/bench_conv.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, torch, time
NO_TORCH = '--no-torch' in sys.argv
sys.path.insert(0, '.')
from jtsm_amd.layers import conv as K
CL = torch.channels_last
cuda = torch.device('cuda:0')
def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [  # N, C, H, W, O, k, s, p
  (2, 4, 1024, 1024, 64, 7, 2, 3),
  (2, 64, 256, 256, 64, 1, 1, 0), (2, 64, 256, 256, 64, 3, 1, 1), (2, 64, 256, 256, 256, 1, 1, 0), (2, 256, 256, 256, 64, 1, 1, 0),
  (2, 256, 256, 256, 128, 1, 2, 0), (2, 128, 128, 128, 128, 3, 1, 1), (2, 128, 128, 128, 512, 1, 1, 0), (2, 512, 128, 128, 128, 1, 1, 0),
  (2, 256, 64, 64, 256, 3, 1, 1), (2, 256, 64, 64, 1024, 1, 1, 0), (2, 1024, 64, 64, 256, 1, 1, 0),
  (2, 512, 32, 32, 512, 3, 1, 1), (2, 512, 32, 32, 2048, 1, 1, 0), (2, 2048, 32, 32, 512, 1, 1, 0),
  (2, 256, 256, 256, 256, 3, 1, 1), (2, 256, 128, 128, 256, 3, 1, 1),
  (4000, 12544, 1, 1, 2048, 1, 1, 0), (4000, 2048, 1, 1, 4096, 1, 1, 0), (4000, 4096, 1, 1, 1870, 1, 1, 0),
]
print("%-44s %9s %9s %9s | torch %9s %9s %9s" % ("shape", "fwd TF", "dgrad TF", "wgrad TF", "fwd", "dgrad", "wgrad"))
for (N, C, H, W, O, k, s, p) in shapes:
    x = torch.randn(N, C, H, W, device=cuda).contiguous(memory_format=CL)
    w = (torch.randn(O, C, k, k, device=cuda) * 0.05).contiguous(memory_format=CL)
    y = K.conv2d_forward(x, w, s, p, 1)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * C * k * k
    t_f = timeit(lambda: K.conv2d_forward(x, w, s, p, 1))
    if O % 4 == 0:
        t_d = timeit(lambda: K.conv2d_backward_data(dy, w, tuple(x.shape), s, p, 1))
        t_w = timeit(lambda: K.conv2d_backward_weight(dy, x, tuple(w.shape), s, p, 1))
    else:
        t_d = t_w = float('nan')
    if NO_TORCH:
        print("%-44s %9.1f %9.1f %9.1f   (ms %.3f %.3f %.3f)" % (str((N, C, H, W, O, k, s, p)), fl/t_f/1e9, fl/t_d/1e9, fl/t_w/1e9, t_f, t_d, t_w), flush=True)
        continue
    # torch baseline
    xr = x.clone().requires_grad_(); wr = w.clone().requires_grad_()
    F = torch.nn.functional
    tt_f = timeit(lambda: F.conv2d(xr, wr, None, s, p))
    yr = F.conv2d(xr, wr, None, s, p)
    tt_d = timeit(lambda: torch.autograd.grad(yr, xr, dy, retain_graph=True))
    tt_w = timeit(lambda: torch.autograd.grad(yr, wr, dy, retain_graph=True))
    print("%-44s %9.1f %9.1f %9.1f | torch %9.1f %9.1f %9.1f   (ms %.3f %.3f %.3f)" % (str((N, C, H, W, O, k, s, p)), fl/t_f/1e9, fl/t_d/1e9, fl/t_w/1e9, fl/tt_f/1e9, fl/tt_d/1e9, fl/tt_w/1e9, t_f, t_d, t_w), flush=True)
