"""Small bandwidth-bound kernels next to the contractions (jtsm_amd/csrc/elementwise.hip)."""
import ctypes as C

import torch

from .. import _lib as L

CL = torch.channels_last


def _dense_like(t, ref):
    """Same storage order as ref (dense)."""
    if ref.dim() == 4 and L.is_nhwc(ref):
        return t.contiguous(memory_format=CL)
    return t.contiguous()


def relu_backward(dy, y, emit_planes=False):
    """g = dy * (y > 0).  emit_planes: also write g's bf16 hi/lo planes (registered for the bf16x3 gradients)."""
    L.require_gpu(dy, y)
    y = y if (y.is_contiguous() or (y.dim() == 4 and y.is_contiguous(memory_format=CL))) else y.contiguous()
    dy = _dense_like(dy, y)
    g = torch.empty_like(y)
    n = y.numel()
    if emit_planes and n % 8 == 0 and n > 0:
        from . import conv
        buf = conv._planes_buf(n, y.device)
        hi, lo = conv._hl(buf)
        L.note_bytes((14.0 if conv.MATH == "f16" else 16.0) * n)   # dy, y read; g and its planes written
        if conv.MATH == "f16":
            L.check(L.lib().jtsm_relu_backward_split_f16(L.ptr(dy), L.ptr(y), L.ptr(g), hi, C.c_long(n), conv.GRAD_SHIFT,
                                                         L.stream()), "relu_backward_split_f16")
        else:
            L.check(L.lib().jtsm_relu_backward_split_f32(L.ptr(dy), L.ptr(y), L.ptr(g), hi, lo, C.c_long(n),
                                                         L.stream()), "relu_backward_split")
        conv.planes_put(g, buf)
        return g
    L.check(L.lib().jtsm_relu_backward_f32(L.ptr(dy), L.ptr(y), L.ptr(g), C.c_long(n), L.stream()),
            "relu_backward")
    return g


def channel_sum(g):
    """Sum over every axis but channels of a (N,C,H,W) channels_last or (R,C) tensor -> (C,)."""
    L.require_gpu(g)
    if g.dim() == 4:
        g = g.contiguous(memory_format=CL)
        rows, ch = g.shape[0] * g.shape[2] * g.shape[3], g.shape[1]
    else:
        g = g.contiguous()
        rows, ch = g.shape
    out = torch.empty(ch, dtype=g.dtype, device=g.device)
    from . import conv
    lib = L.lib()
    nbytes = lib.jtsm_channel_sum_workspace_bytes(C.c_long(rows), ch)
    ws = conv._scratch(nbytes, g.device)      # the contractions' scratch: same stream, in order
    L.note_bytes(4.0 * g.numel())
    L.check(lib.jtsm_channel_sum_ws_f32(L.ptr(g), L.ptr(out), C.c_long(rows), ch, L.ptr(ws), C.c_size_t(nbytes),
                                        L.stream()), "channel_sum")
    return out


def relu_backward_scaled(dy, y, scale):
    """(g, planes buffer of g): g = y > 0 ? dy * scale : 0 — the backward of ReLU followed by inverted dropout when y is
    the dropout's output and scale = 1 / (1 - p) (scale 1: a plain ReLU gate).  Needs numel % 8 == 0."""
    from . import conv
    L.require_gpu(dy, y)
    y = y.contiguous()
    dy = dy.contiguous()
    g = torch.empty_like(y)
    n = y.numel()
    buf = conv._planes_buf(n, y.device)
    hi, lo = conv._hl(buf)
    L.note_bytes((14.0 if conv.MATH == "f16" else 16.0) * n)
    L.check(L.lib().jtsm_relu_backward_split_scaled_f32(
        L.ptr(dy), L.ptr(y), L.f32(scale), L.ptr(g), hi, lo, C.c_long(n), conv.GRAD_SHIFT if conv.MATH == "f16" else 0,
        L.stream()), "relu_backward_split_scaled")
    return g, buf


def split_rowscale(x2d, row_scale):
    """Planes buffer of x2d[r][c] * row_scale[r] (x2d dense (R, K), K % 8 == 0); the product itself is never stored."""
    from . import conv
    L.require_gpu(x2d, row_scale)
    r, k = x2d.shape
    buf = conv._planes_buf(r * k, x2d.device)
    hi, lo = conv._hl(buf)
    L.note_bytes((6.0 if conv.MATH == "f16" else 8.0) * r * k)
    L.check(L.lib().jtsm_split_rowscale_f32(L.ptr(x2d), L.ptr(row_scale), C.c_long(r), k, hi, lo, 0, L.stream()),
            "split_rowscale")
    return buf


def dropout_split_(y, p, seed):
    """In place: y <- inverted dropout of y with the counter-based mask of `seed`; returns the planes buffer of the
    result (numel % 8 == 0)."""
    from . import conv
    L.require_gpu(y)
    n = y.numel()
    buf = conv._planes_buf(n, y.device)
    hi, lo = conv._hl(buf)
    L.note_bytes((10.0 if conv.MATH == "f16" else 12.0) * n)
    L.check(L.lib().jtsm_dropout_split_f32(L.ptr(y), L.ptr(y), hi, lo, C.c_long(n), L.f32(p), C.c_ulonglong(seed),
                                           L.stream()), "dropout_split")
    return buf


def _cl4(x):
    if x.dim() != 4 or x.shape[1] % 4:
        raise RuntimeError("jtsm_amd spatial helpers need a (N,C,H,W) tensor with C %% 4 == 0, got %s" % (tuple(x.shape),))
    L.require_gpu(x)
    return x.contiguous(memory_format=CL)


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl4(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=CL)
        L.check(L.lib().jtsm_maxpool3x3s2_forward_f32(L.ptr(x), L.ptr(y), n, h, w, c, L.stream()), "maxpool")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        gy = gy.contiguous(memory_format=CL)
        gx = torch.empty_like(x)
        L.check(L.lib().jtsm_maxpool3x3s2_backward_f32(L.ptr(x), L.ptr(gy), L.ptr(gx), n, h, w, c, L.stream()),
                "maxpool backward")
        return gx


class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        x = _cl4(x)
        n, c, h, w = x.shape
        ho, wo = (h, w) if stride == 1 else (h // 2, w // 2)
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=CL)
        L.check(L.lib().jtsm_maxpool2x2_forward_f32(L.ptr(x), L.ptr(y), n, h, w, c, stride, L.stream()), "maxpool2x2")
        ctx.save_for_backward(x)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        gy = gy.contiguous(memory_format=CL)
        gx = torch.empty_like(x)
        L.check(L.lib().jtsm_maxpool2x2_backward_f32(L.ptr(x), L.ptr(gy), L.ptr(gx), n, h, w, c, ctx.stride,
                                                     L.stream()), "maxpool2x2 backward")
        return gx, None


def max_pool_2x2(x, stride=2):
    """stride 2: F.max_pool2d(x, 2, 2); stride 1: F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1) — the two pooling
    forms of the WSL ResNet-v2 backbone — on a channels_last tensor."""
    if stride not in (1, 2):
        raise ValueError("max_pool_2x2: stride must be 1 or 2")
    return _MaxPool2x2.apply(x, stride)


def max_pool_3x3_s2(x):
    """F.max_pool2d(x, kernel_size=3, stride=2, padding=1) on a channels_last tensor."""
    return _MaxPool3s2.apply(x)


class _Upsample2Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, top, lateral, fan=None):
        ctx.fan = fan   # (layers/grad_fan.py: upsample2_add claimed `top`'s fan view, if it is one)
        top, lateral = _cl4(top), _cl4(lateral)
        n, c, h, w = lateral.shape
        if top.shape != (n, c, h // 2, w // 2) or h % 2 or w % 2:
            raise RuntimeError("upsample2_add: lateral %s is not 2x top %s" % (tuple(lateral.shape), tuple(top.shape)))
        out = torch.empty_like(lateral)
        from . import conv
        buf = hi = lo = None
        if conv.MATH == "bf16x3" and c % 8 == 0 and out.numel() > 0:   # the merged map feeds a 3x3 output conv
            buf = conv._planes_buf(out.numel(), out.device)
            hi, lo = conv._hl(buf)
        L.check(L.lib().jtsm_upsample2_add_f32(L.ptr(top), L.ptr(lateral), L.ptr(out), hi, lo, n, h, w, c, L.stream()),
                "upsample2_add")
        if buf is not None:
            conv.planes_put(out, buf)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous(memory_format=CL)
        n, c, h, w = g.shape
        gt = None
        if ctx.needs_input_grad[0]:
            gt = torch.empty((n, c, h // 2, w // 2), dtype=g.dtype, device=g.device, memory_format=CL)
            L.check(L.lib().jtsm_sum2x2_f32(L.ptr(g), L.ptr(gt), n, h // 2, w // 2, c, L.stream()), "sum2x2")
            # `top` (an FPN merged map) is also read by its level's output convolution, which runs backward after this
            # node and adds into this map (layers/grad_fan.py); were a map there already, autograd adds the two
            from . import grad_fan
            if grad_fan.target(ctx.fan, gt.shape, gt.device) is None and grad_fan.offer(ctx.fan, gt):
                gt = None
        return gt, (g if ctx.needs_input_grad[1] else None), None


class _SumTensors(torch.autograd.Function):
    """x0 + x1 + ... (list order) in one pass, with the bf16 planes of the sum; every input's gradient is the
    incoming gradient itself."""

    @staticmethod
    def forward(ctx, *xs):
        from . import conv
        xs = [_cl4(x) for x in xs]
        out = torch.empty_like(xs[0])
        n = out.numel()
        buf = hi = lo = None
        if conv.MATH == "bf16x3" and out.shape[1] % 8 == 0 and n > 0:
            buf = conv._planes_buf(n, out.device)
            hi, lo = conv._hl(buf)
        ptrs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        L.check(L.lib().jtsm_sum_tensors_f32(ptrs, len(xs), C.c_long(n), L.ptr(out), hi, lo, L.stream()), "sum_tensors")
        if buf is not None:
            conv.planes_put(out, buf)
        ctx.n = len(xs)
        return out

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


def sum_tensors(xs):
    """Sum of 1..4 same-shape channels_last maps (channels % 4 == 0), accumulated in list order."""
    xs = list(xs)
    if len(xs) == 1:
        return xs[0]
    if len(xs) > 4 or any(x.shape != xs[0].shape for x in xs) or xs[0].shape[1] % 4 or not xs[0].is_cuda:
        out = xs[0]
        for x in xs[1:]:
            out = out + x
        return out
    return _SumTensors.apply(*xs)


def upsample2_add(top, lateral):
    """lateral + F.interpolate(top, scale_factor=2, mode="nearest")."""
    from . import grad_fan
    return _Upsample2Add.apply(top, lateral, grad_fan.claim(top))


class _Subsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl4(x)
        n, c, h, w = x.shape
        ctx.shape = (n, c, h, w)
        y = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=CL)
        L.check(L.lib().jtsm_subsample2_f32(L.ptr(x), L.ptr(y), n, h, w, c, 0, L.stream()), "subsample2")
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = g.contiguous(memory_format=CL)
        gx = torch.empty((n, c, h, w), dtype=g.dtype, device=g.device, memory_format=CL)
        L.check(L.lib().jtsm_subsample2_f32(L.ptr(g), L.ptr(gx), n, h, w, c, 1, L.stream()), "subsample2 scatter")
        return gx


def subsample2(x):
    """F.max_pool2d(x, kernel_size=1, stride=2, padding=0)."""
    return _Subsample2.apply(x)


class _GroupNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        x = _cl4(x)
        n, c, h, w = x.shape
        lib = L.lib()
        y = torch.empty_like(x)
        mean = torch.empty((n, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(lib.jtsm_group_norm_workspace_bytes(n, C.c_long(h * w), c), dtype=torch.uint8, device=x.device)
        L.check(lib.jtsm_group_norm_forward_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean),
                                                L.ptr(rstd), L.ptr(ws), n, C.c_long(h * w), c, groups, L.f32(eps),
                                                int(relu), L.stream()), "group_norm_forward")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (groups, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, relu = ctx.cfg
        n, c, h, w = x.shape
        dy = dy.contiguous(memory_format=CL)
        lib = L.lib()
        from . import conv
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(lib.jtsm_group_norm_workspace_bytes(n, C.c_long(h * w), c), dtype=torch.uint8, device=x.device)
        # dx is the output gradient of the convolution in front of this norm: hand it its bf16 planes as well
        buf = conv._planes_buf(dx.numel(), dx.device) if (conv.MATH == "bf16x3" and c % 8 == 0) else None
        hi, lo = conv._hl(buf)
        L.check(lib.jtsm_group_norm_backward_f32(L.ptr(x), L.ptr(dy), L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                                 L.ptr(rstd), L.ptr(dx), hi, lo, L.ptr(dg), L.ptr(db), L.ptr(ws), n,
                                                 C.c_long(h * w), c, groups, int(relu), L.stream()),
                "group_norm_backward")
        if buf is not None:
            conv.planes_put(dx, buf)
        return dx, dg, db, None, None, None


def group_norm_relu(x, gamma, beta, groups, eps=1e-5, relu=True):
    """F.relu(F.group_norm(x, groups, gamma, beta, eps)) on a channels_last tensor, one fused pass."""
    return _GroupNormReLU.apply(x, gamma, beta, groups, eps, relu)


class _UpBilinear2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl4(x)
        n, c, h, w = x.shape
        from . import conv
        y = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=CL)
        # the up-sampled map feeds the next convolution of the sem-seg head: emit its bf16 planes in the same pass
        buf = conv._planes_buf(y.numel(), y.device) if (conv.MATH == "bf16x3" and c % 8 == 0) else None
        hi, lo = conv._hl(buf)
        L.check(L.lib().jtsm_upsample_bilinear2x_forward_f32(L.ptr(x), L.ptr(y), hi, lo, n, h, w, c, L.stream()),
                "upsample2x")
        if buf is not None:
            conv.planes_put(y, buf)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        n, c, h, w = ctx.shape
        gy = gy.contiguous(memory_format=CL)
        gx = torch.empty((n, c, h, w), dtype=gy.dtype, device=gy.device, memory_format=CL)
        L.check(L.lib().jtsm_upsample_bilinear2x_backward_f32(L.ptr(gy), L.ptr(gx), n, h, w, c, L.stream()),
                "upsample2x backward")
        return gx


def upsample_bilinear2x(x):
    """F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), channels_last."""
    return _UpBilinear2x.apply(x)


class _SemSegCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, scale, ignore_index):
        L.require_gpu(logits, target)
        n, c, hs, ws = logits.shape
        # channels-last storage with an arbitrary channel pitch (a [:, :54] slice of a 56-wide map is fine)
        if not (logits.stride(1) == 1 and logits.stride(3) >= c and logits.stride(2) == ws * logits.stride(3)
                and logits.stride(0) == hs * logits.stride(2)):
            logits = logits.contiguous(memory_format=CL)
        if logits.stride(3) % 4 or logits.data_ptr() % 16:
            # the kernels read float4 rows: re-pitch to a multiple of 4 channels (padding is never read as a class)
            ld4 = (c + 3) // 4 * 4
            padded = torch.zeros((n, hs, ws, ld4), dtype=logits.dtype, device=logits.device)
            padded[..., :c] = logits.permute(0, 2, 3, 1)
            logits = padded.permute(0, 3, 1, 2)[:, :c]
        ld = logits.stride(3)
        target = target.to(torch.int64).contiguous()
        if tuple(target.shape) != (n, hs * scale, ws * scale):
            raise RuntimeError("semseg_ce: target %s is not %dx the logits %s" % (tuple(target.shape), scale,
                                                                                   tuple(logits.shape)))
        lib = L.lib()
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        wsb = torch.empty(lib.jtsm_semseg_ce_workspace_bytes(n, hs, ws, scale), dtype=torch.uint8, device=logits.device)
        L.note_bytes(4.0 * n * hs * ws * c + 8.0 * target.numel())     # stride-4 logits + full-resolution labels
        L.check(lib.jtsm_semseg_ce_forward_f32(L.ptr(logits), ld, c, L.ptr(target), L.ptr(out), L.ptr(wsb), n, hs, ws,
                                               scale, C.c_long(ignore_index), L.stream()), "semseg_ce_forward")
        ctx.save_for_backward(logits, target, out, wsb)
        ctx.cfg = (ld, scale, ignore_index)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, out, wsb = ctx.saved_tensors
        ld, scale, ignore_index = ctx.cfg
        n, c, hs, ws = logits.shape
        dfull = torch.empty((n, hs, ws, ld), dtype=torch.float32, device=logits.device)
        g = g.to(torch.float32).contiguous()
        L.note_bytes(4.0 * n * hs * ws * c + 8.0 * target.numel() + 4.0 * dfull.numel())
        L.check(L.lib().jtsm_semseg_ce_backward_f32(L.ptr(logits), ld, c, L.ptr(target), L.ptr(out), L.ptr(g),
                                                    L.ptr(dfull), L.ptr(wsb), n, hs, ws, scale,
                                                    C.c_long(ignore_index), L.stream()), "semseg_ce_backward")
        if ld != c:
            from .conv import ZERO_PADDED
            setattr(dfull, ZERO_PADDED, True)   # (the kernel wrote zeros into channels c .. ld - 1)
        return dfull.permute(0, 3, 1, 2)[:, :c], None, None, None


def semseg_cross_entropy(logits, target, scale=4, ignore_index=255):
    """F.cross_entropy(F.interpolate(logits, scale_factor=scale, mode="bilinear", align_corners=False),
    target, reduction="mean", ignore_index=ignore_index) without materialising the up-sampled logits."""
    return _SemSegCE.apply(logits, target, int(scale), int(ignore_index))
