"""Per-layer effect of fp16-only outputs: conv3 of a bottleneck (1x1, C/4 -> C, residual, ReLU) and conv1's data gradient
(1x1, with the shortcut term) in the fp16 arithmetic — fp32 result + plane + fp32 residual against plane only + fp16
residual plane, hipEvent timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from jtsm_amd.layers import conv as K

dev = torch.device("cuda", 0)
CL = torch.channels_last
K.set_math("f16")

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (B, H, W, C) in ((2, 256, 512, 256), (2, 128, 256, 512), (2, 64, 128, 1024), (2, 32, 64, 2048), (2, 64, 64, 1024)):
    mid = C // 4
    y2 = K.PlaneTensor.of(torch.relu(torch.randn(B, mid, H, W, device=dev)).contiguous(memory_format=CL))
    x = torch.relu(torch.randn(B, C, H, W, device=dev)).contiguous(memory_format=CL)
    xp = K.PlaneTensor.of(x)
    w3 = (torch.randn(C, mid, 1, 1, device=dev) * 0.05).contiguous(memory_format=CL)
    w1 = (torch.randn(mid, C, 1, 1, device=dev) * 0.05).contiguous(memory_format=CL)
    s3, b3 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    s1 = torch.rand(mid, device=dev) + 0.5
    t_a = timeit(lambda: K.planes_forward(y2, w3, 1, 0, 1, b3, True, fp32="both", scale=s3, residual=x))
    t_b = timeit(lambda: K.planes_forward(y2, w3, 1, 0, 1, b3, True, scale=s3, residual_plane=xp))
    g = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL)
    gp = K.PlaneTensor.of(g, grad=True)
    d1 = K.PlaneTensor.of(torch.randn(B, mid, H, W, device=dev).contiguous(memory_format=CL), grad=True)
    t_c = timeit(lambda: K.planes_backward_data(d1, w1, x.shape, 1, 0, 1, both=True, accumulate=g, kscale=s1, gate=xp))
    t_d = timeit(lambda: K.planes_backward_data(d1, w1, x.shape, 1, 0, 1, kscale=s1, accumulate_plane=gp, gate=xp))
    n = B * H * W * C
    print("%dx%dx%dx%d: conv3 fwd %.1f -> %.1f us (%.0f -> %.0f MB); conv1 dgrad %.1f -> %.1f us" % (
        B, H, W, C, t_a, t_b, n * (4 + 2 + 4 + 0.5) / 1e6, n * (2 + 2 + 0.5) / 1e6, t_c, t_d), flush=True)
