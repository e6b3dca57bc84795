"""Is the 256 x 256-tile kernel bound inside the CU or by what the CUs share?  The same per-workgroup work on 256, 128,
64 and 32 workgroups (one per CU): equal times = bound inside the CU."""
import sys, torch
sys.path.insert(0, '.')
from jtsm_amd.layers import conv as K
cuda = torch.device('cuda:0')
CL = torch.channels_last
def kernel_ms(f, n=10):
    K.LAUNCH_LOG = []
    for _ in range(n): f()
    torch.cuda.synchronize()
    t = sorted(sp.kernel_ms() for (_, _, sp, _, _) in K.LAUNCH_LOG)
    v = K.LAUNCH_LOG[0][0]
    K.LAUNCH_LOG = None
    return t[len(t) // 2], v
for (N, C, O) in [(4096, 6272, 4096), (4096, 6272, 2048), (2048, 6272, 2048), (2048, 6272, 1024), (1024, 6272, 1024)]:
    x = torch.randn(N, C, 1, 1, device=cuda).contiguous(memory_format=CL)
    w = (torch.randn(O, C, 1, 1, device=cuda) * 0.05).contiguous(memory_format=CL)
    a, v = kernel_ms(lambda: K.conv2d_forward(x, w, 1, 0, 1))
    fl = 2.0 * N * O * C
    print("M %5d N %5d K %5d: %3d tiles, %s splits %s  %.3f ms  %.0f TF/s  (%.1f TF/s per tile-CU)" % (
        N, O, C, (N // 256) * (O // 256), str(v), getattr(v, "splits", 1), a, fl / a / 1e9, fl / a / 1e9 / ((N // 256) * (O // 256))), flush=True)
