"""A whole ResNet bottleneck (detectron2/modeling/backbone/resnet.py:101-211 with FrozenBN) as ONE autograd node.

Forward is the same three or four fused convolution launches as the layer-by-layer form.  The point is the
backward: inside one node the gradient chain can use the data-gradient kernel's epilogue for everything that
sits between two contractions —
    * the ReLU gates of conv1 / conv2 outputs (`relu_mask`): no separate relu_backward pass, and the gated
      gradient leaves the kernel together with its bf16 planes, ready for the next two contractions;
    * the sum of the two gradient paths into the block input (`accumulate`): no autograd add.
Per block that removes two elementwise passes over the bottleneck activations and one over the block input.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import conv as K
from . import grad_fan
from .elementwise import relu_backward

CL = torch.channels_last
# False: residual blocks run layer by layer (the same kernels, one autograd node per convolution) — used by the
# frozen-gates parity test, whose forward hooks need every convolution's output
ENABLED = True
MASK_TOWER = os.environ.get("JTSM_MASK_TOWER", "1") != "0"   # (A/B switch for the mask-head node alone)
PREGATE = os.environ.get("JTSM_BLOCK_PREGATE", "1") != "0"   # (A/B switch: block-output gates in the next block's epilogue)
BIAS_COLSUM = os.environ.get("JTSM_BIAS_COLSUM", "1") != "0"  # (A/B switch: mask-tower bias gradients in the data-gradient epilogues)


def _dgrad(g, w, scale, x_shape, stride, pad, dil, accumulate=None, relu_mask=None, emit_planes=False):
    """Data gradient through conv + FrozenBN scale (folded into the weight rows one way or the other)."""
    if K.MATH != "f32":
        return K.conv2d_backward_data(g, w, x_shape, stride, pad, dil, kscale=scale, accumulate=accumulate,
                                      relu_mask=relu_mask, emit_planes=emit_planes)
    w_eff = w if scale is None else (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
    return K.conv2d_backward_data(g, w_eff, x_shape, stride, pad, dil, accumulate=accumulate, relu_mask=relu_mask)


# The last data gradient an identity block handed back ALREADY gated by its input's ReLU: (tensor, version).  Inside a
# stage block k's input is block k-1's ReLU output, so the gate of block k-1's output gradient can ride in block k's
# conv1 data-gradient epilogue (accumulate the shortcut path, gate, emit planes) instead of a separate
# relu_backward pass over the stage's widest activation.  The gate is idempotent and linear, so handing a gated
# term to a consumer that gates again is always correct; SKIPPING the second gate is correct only if the gradient
# block k-1 receives is exactly that tensor.  Holding the tensor here keeps autograd from accumulating another
# consumer's term into it in place (it adds into a fresh tensor instead, which then fails the identity check below
# and is gated the ordinary way).
_PREGATED = [None]
# fp16 arithmetic: the identity-shortcut blocks of a stage as one node on fp16 planes only (JTSM_CHAIN16=0: per-block nodes)
CHAIN16 = os.environ.get("JTSM_CHAIN16", "1") != "0"


def _drop_pregated():
    """A gradient handed over but never taken (the previous block frozen or detached) must not stay pinned — a whole
    res-stage activation gradient — beyond the step: cleared with the plane cache, at the start of every forward."""
    _PREGATED[0] = None


K.CLEAR_HOOKS.append(_drop_pregated)


def _hand_pregated(dx):
    _PREGATED[0] = (dx, dx._version)


def _take_pregated(dy):
    m = _PREGATED[0]
    if m is None:
        return False
    _PREGATED[0] = None
    return m[0].data_ptr() == dy.data_ptr() and m[0].shape == dy.shape and m[1] == dy._version


def _same_strides(dw, w):
    if dw is not None and dw.stride() != w.stride() and w.shape[2] == 1 and w.shape[3] == 1:
        return dw.as_strided(w.shape, w.stride())
    return dw


class _BottleneckFn(Function):
    """In the plane arithmetics the two inner activations (conv1 / conv2 outputs) exist as operand planes only
    (layers/conv.py: PlaneTensor): no fp32 copy is written, their ReLU gates are read from the hi plane in the
    data-gradient epilogues.  The block input and output stay fp32 (the residual stream keeps full precision)."""

    @staticmethod
    def forward(ctx, x, w1, s1, b1, w2, s2, b2, w3, s3, b3, ws, ss, bs, stride1, stride2, pad2, dil2, stride_s,
                pregate=False, fan=None):
        ctx.cfg = (stride1, stride2, pad2, dil2, stride_s)
        ctx.segment = K.SEGMENT
        ctx.fan = fan   # (layers/grad_fan.py: bottleneck_fused claimed x's fan view, if it is one)
        # x is the ReLU output of the previous block's node (bottleneck_fused tags it): its gate goes into this
        # block's conv1 data-gradient epilogue
        ctx.pregate = bool(pregate) and ws is None
        ctx.planes = K.MATH != "f32" and _plane_block_ok(x, w1, w2, w3, ws)
        if ctx.planes:
            xp = K.PlaneTensor.of(x)
            y1 = K.planes_forward(xp, w1, stride1, 0, 1, b1, True, scale=s1)
            y2 = K.planes_forward(y1, w2, stride2, pad2, dil2, b2, True, scale=s2)
            sc = x if ws is None else K.planes_forward(xp, ws, stride_s, 0, 1, bs, False, fp32=True, scale=ss)
            y3, y3p = K.planes_forward(y2, w3, 1, 0, 1, b3, True, fp32="both", scale=s3, residual=sc)
            K.planes_put(y3, y3p.buf)
            ctx.inner = (xp, y1, y2)
            ctx.save_for_backward(x, y3, w1, s1, w2, s2, w3, s3, ws, ss)
            return y3
        y1 = K.conv2d_forward(x, w1, stride1, 0, 1, s1, b1, None, True, emit_planes=True)
        y2 = K.conv2d_forward(y1, w2, stride2, pad2, dil2, s2, b2, None, True, emit_planes=True)
        sc = x if ws is None else K.conv2d_forward(x, ws, stride_s, 0, 1, ss, bs, None, False)
        y3 = K.conv2d_forward(y2, w3, 1, 0, 1, s3, b3, sc, True, emit_planes=True)
        ctx.save_for_backward(x, y1, y2, y3, w1, s1, w2, s2, w3, s3, ws, ss)
        return y3

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        K.set_segment(ctx.segment)
        if ctx.planes:
            return _BottleneckFn._backward_planes(ctx, dy)
        x, y1, y2, y3, w1, s1, w2, s2, w3, s3, ws, ss = ctx.saved_tensors
        stride1, stride2, pad2, dil2, stride_s = ctx.cfg
        need = ctx.needs_input_grad
        x3 = K.MATH != "f32"
        dx = dw1 = dw2 = dw3 = dws = None
        # the block's own output gate (already applied by the next block's conv1 data gradient inside a stage)
        g3 = dy if _take_pregated(dy) else relu_backward(dy, y3, emit_planes=x3)
        if need[7]:
            dw3 = K.conv2d_backward_weight(g3, y2, tuple(w3.shape), 1, 0, 1, row_scale=s3, w=w3)
        # gradient at conv2's output, gated by its ReLU in the epilogue
        d2 = _dgrad(g3, w3, s3, tuple(y2.shape), 1, 0, 1, relu_mask=y2, emit_planes=True)
        if need[4]:
            dw2 = K.conv2d_backward_weight(d2, y1, tuple(w2.shape), stride2, pad2, dil2, row_scale=s2, w=w2)
        d1 = _dgrad(d2, w2, s2, tuple(y1.shape), stride2, pad2, dil2, relu_mask=y1, emit_planes=True)
        if need[1]:
            dw1 = K.conv2d_backward_weight(d1, x, tuple(w1.shape), stride1, 0, 1, row_scale=s1, w=w1)
        if ws is not None and need[10]:
            dws = K.conv2d_backward_weight(g3, x, tuple(ws.shape), stride_s, 0, 1, row_scale=ss, w=ws)
        if need[0]:
            xs = tuple(x.shape)
            if ws is None:
                # identity shortcut: the block input receives g3 directly, added in conv1's data-gradient epilogue
                if ctx.pregate:
                    dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1, accumulate=g3, relu_mask=x, emit_planes=x3)
                    _hand_pregated(dx)
                else:
                    dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1, accumulate=g3)
            elif stride1 == 1 and stride_s == 1:
                dxs = _dgrad(g3, ws, ss, xs, stride_s, 0, 1)
                dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1, accumulate=dxs)
            else:
                # strided 1x1 pair (first block of a stage): both run as dense GEMM + scatter, summed once
                dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1)
                dx = dx.add_(_dgrad(g3, ws, ss, xs, stride_s, 0, 1))
        return (dx, _same_strides(dw1, w1), None, None, _same_strides(dw2, w2), None, None, _same_strides(dw3, w3),
                None, None, _same_strides(dws, ws) if ws is not None else None, None, None, None, None, None, None, None,
                None, None)

    @staticmethod
    def _backward_planes(ctx, dy):
        x, y3, w1, s1, w2, s2, w3, s3, ws, ss = ctx.saved_tensors
        xp, y1, y2 = ctx.inner
        stride1, stride2, pad2, dil2, stride_s = ctx.cfg
        need = ctx.needs_input_grad
        dx = dw1 = dw2 = dw3 = dws = None
        # the block's own output gate (fp32 + planes) — inside a stage the next block's conv1 data gradient applied it
        g3 = dy if _take_pregated(dy) else relu_backward(dy, y3, emit_planes=True)
        g3p = K.PlaneTensor.of(g3, grad=True)
        # (weight gradients: queued for the grouped launch of the stage's same-shape layers, layers/conv.py — a queued
        # gradient reaches the parameter without passing through autograd, so the node returns None for it)
        if need[7]:
            dw3 = K.planes_backward_weight_deferred(g3p, y2, w3, 1, 0, 1, row_scale=s3)
        d2 = K.planes_backward_data(g3p, w3, y2.shape, 1, 0, 1, gate=y2, kscale=s3)      # gated by conv2's ReLU
        if need[4]:
            dw2 = K.planes_backward_weight_deferred(d2, y1, w2, stride2, pad2, dil2, row_scale=s2)
        d1 = K.planes_backward_data(d2, w2, y1.shape, stride2, pad2, dil2, gate=y1, kscale=s2)
        if need[1]:
            dw1 = K.planes_backward_weight_deferred(d1, xp, w1, stride1, 0, 1, row_scale=s1)
        if ws is not None and need[10]:
            dws = K.planes_backward_weight_deferred(g3p, xp, ws, stride_s, 0, 1, row_scale=ss)
        if need[0]:
            xs = tuple(x.shape)
            if ws is None:
                if ctx.pregate:
                    dx, dxp = K.planes_backward_data(d1, w1, xs, stride1, 0, 1, both=True, accumulate=g3, kscale=s1,
                                                     gate=xp)
                    K.planes_put(dx, dxp.buf)
                    _hand_pregated(dx)
                else:
                    dx = K.planes_backward_data(d1, w1, xs, stride1, 0, 1, fp32=True, accumulate=g3, kscale=s1)
            else:
                # the input of a stage's first block is a res-stage output, which the FPN lateral reads too: when its
                # gradient map exists already (layers/grad_fan.py) both paths add into it, and so does the second path
                # into the first path's map otherwise — no separate addition either way
                sink = grad_fan.target(ctx.fan, xs, dy.device)
                if sink is None:
                    dx = K.planes_backward_data(g3p, ws, xs, stride_s, 0, 1, fp32=True, kscale=ss)
                else:
                    K.planes_backward_data(g3p, ws, xs, stride_s, 0, 1, fp32=True, kscale=ss, into=sink)
                K.planes_backward_data(d1, w1, xs, stride1, 0, 1, fp32=True, kscale=s1, into=sink if sink is not None else dx)
                if sink is not None or grad_fan.offer(ctx.fan, dx):
                    dx = None
        if ws is not None:   # the first block of a stage is the last of it to run backward: launch the stage's groups
            K.flush_deferred_weight_gradients()
        return (dx, _same_strides(dw1, w1), None, None, _same_strides(dw2, w2), None, None, _same_strides(dw3, w3),
                None, None, _same_strides(dws, ws) if ws is not None else None, None, None, None, None, None, None, None,
                None, None)


class _IdentityChain16Fn(Function):
    """fp16-ONLY activations (BASELINE configs[4]; the reference's AMP step keeps activations and their gradients in
    fp16, detectron2/engine/train_loop.py:289-336): the identity-shortcut blocks of a stage — every block after its
    first — as ONE autograd node in the fp16 arithmetic.  Inside it an activation exists as its fp16 operand plane and
    nothing else: conv3's epilogue reads the residual from the block input's plane (jtsm_conv2d_forward_res16_f16) and
    writes the block output's plane only; backward, the gradient stream between the blocks is an fp16 plane carrying
    2^GRAD_SHIFT — conv1's data-gradient epilogue adds the shortcut term from the previous plane
    (jtsm_conv2d_backward_data_acc16_f16), gates by the block input's plane and emits the next one.  fp32 exists for the
    chain's input (the first block's output, whose plane is reused), its output (read by the FPN lateral and the next
    stage) and the two gradients at those ends.  Per inner block output that is 2 B written + 2 B read instead of
    4 + 2 B written + 4 B read.

    apply(x, pad2, dil2, n, w1, s1, b1, w2, s2, b2, w3, s3, b3, ... per block) -> y."""

    @staticmethod
    def forward(ctx, x, pad2, dil2, n, *params):
        ctx.segment = K.SEGMENT
        inp = K.PlaneTensor.of(x)
        saved, y = [], None
        for k in range(n):
            w1, s1, b1, w2, s2, b2, w3, s3, b3 = params[9 * k:9 * k + 9]
            y1 = K.planes_forward(inp, w1, 1, 0, 1, b1, True, scale=s1)
            y2 = K.planes_forward(y1, w2, 1, pad2, dil2, b2, True, scale=s2)
            if k == n - 1:
                y, yp = K.planes_forward(y2, w3, 1, 0, 1, b3, True, fp32="both", scale=s3, residual_plane=inp)
                K.planes_put(y, yp.buf)
            else:
                yp = K.planes_forward(y2, w3, 1, 0, 1, b3, True, scale=s3, residual_plane=inp)
            saved.append((inp, y1, y2))
            inp = yp
        ctx.cfg, ctx.inner, ctx.xshape = (pad2, dil2, n), saved, tuple(x.shape)
        ctx.save_for_backward(y, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        K.set_segment(ctx.segment)
        pad2, dil2, n = ctx.cfg
        y, params = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        need = ctx.needs_input_grad
        grads = [None] * (9 * n)
        # the chain's own output gate (the next stage's first block never pre-gates: it has a projection shortcut)
        g = dy if _take_pregated(dy) else relu_backward(dy, y, emit_planes=True)
        gp = K.PlaneTensor.of(g, grad=True)
        dx = None
        for k in range(n - 1, -1, -1):
            w1, s1, b1, w2, s2, b2, w3, s3, b3 = params[9 * k:9 * k + 9]
            inp, y1, y2 = ctx.inner[k]
            base = 4 + 9 * k
            if need[base + 6]:
                grads[9 * k + 6] = _same_strides(K.planes_backward_weight_deferred(gp, y2, w3, 1, 0, 1, row_scale=s3), w3)
            d2 = K.planes_backward_data(gp, w3, y2.shape, 1, 0, 1, gate=y2, kscale=s3)
            if need[base + 3]:
                grads[9 * k + 3] = _same_strides(K.planes_backward_weight_deferred(d2, y1, w2, 1, pad2, dil2, row_scale=s2), w2)
            d1 = K.planes_backward_data(d2, w2, y1.shape, 1, pad2, dil2, gate=y1, kscale=s2)
            if need[base]:
                grads[9 * k] = _same_strides(K.planes_backward_weight_deferred(d1, inp, w1, 1, 0, 1, row_scale=s1), w1)
            if k > 0:
                # gradient at this block's input = the previous block's output: conv1's term + the shortcut's (gp), gated
                # by that output's ReLU (its plane) — a plane only
                gp = K.planes_backward_data(d1, w1, inp.shape, 1, 0, 1, kscale=s1, accumulate_plane=gp, gate=inp)
            elif need[0]:
                # the chain's input is the first block's ReLU output: its gate rides here too, and the first block's
                # node takes the gradient as already gated (fp32 + planes)
                dx, dxp = K.planes_backward_data(d1, w1, ctx.xshape, 1, 0, 1, kscale=s1, accumulate_plane=gp, gate=inp,
                                                 both=True)
                K.planes_put(dx, dxp.buf)
                _hand_pregated(dx)
        ctx.inner = None
        return (dx, None, None, None) + tuple(grads)


def identity_chain_ok(x, blocks):
    """The blocks after a stage's first one as an fp16-only chain: fp16 arithmetic, fused nodes on, frozen norms,
    identity shortcuts, stride 1, one 3x3 geometry, plane-eligible widths."""
    if not (ENABLED and CHAIN16 and K.MATH == "f16" and blocks and x.is_cuda and x.dtype == torch.float32 and
            x.shape[0] > 0):
        return False
    pad2, dil2 = blocks[0].conv2.padding[0], blocks[0].conv2.dilation[0]
    for b in blocks:
        if getattr(b, "shortcut", True) is not None:
            return False
        cs = (b.conv1, b.conv2, b.conv3)
        if any(c.stride[0] != 1 for c in cs) or b.conv2.padding[0] != pad2 or b.conv2.dilation[0] != dil2:
            return False
        if any(not hasattr(c.norm, "scale_bias") for c in cs):
            return False
        if not _plane_block_ok(x, b.conv1.weight, b.conv2.weight, b.conv3.weight, None):
            return False
    return True


def identity_chain_fused(x, blocks):
    params = []
    for b in blocks:
        for c in (b.conv1, b.conv2, b.conv3):
            s, bi = c.norm.scale_bias()
            params += [c.weight, s, bi]
    y = _IdentityChain16Fn.apply(x, blocks[0].conv2.padding[0], blocks[0].conv2.dilation[0], len(blocks), *params)
    y._jtsm_block_relu_out = True
    return y


def _plane_block_ok(x, w1, w2, w3, ws):
    """Every contraction of the block eligible for the plane kernels in all three roles (whole 32-channel K stages,
    16-byte rows), a non-empty batch."""
    if x.shape[0] == 0 or not x.is_cuda:
        return False
    for w in (w1, w2, w3, ws):
        if w is not None and (w.shape[0] % 32 or w.shape[1] % 32):
            return False
    return True


def bottleneck_fused(x, w1, sb1, w2, sb2, w3, sb3, ws, sbs, stride1, stride2, pad2, dil2, stride_s):
    ss, bs = sbs if sbs is not None else (None, None)
    pregate = PREGATE and ws is None and getattr(x, "_jtsm_block_relu_out", False)
    fan = grad_fan.claim(x) if ws is not None else None
    y = _BottleneckFn.apply(x, w1, sb1[0], sb1[1], w2, sb2[0], sb2[1], w3, sb3[0], sb3[1], ws, ss, bs,
                            stride1, stride2, pad2, dil2, stride_s, pregate, fan)
    y._jtsm_block_relu_out = True    # (a tag on this Python object: any op in between yields an untagged tensor)
    return y


class _MaskTowerFn(Function):
    """The mask head's layers (detectron2/modeling/roi_heads/mask_head.py:201-290; WSL variant
    projects/WSL/wsl/modeling/roi_heads/mask_head.py:266-343) as ONE autograd node: k x [conv3x3 + bias + ReLU],
    ConvTranspose2d(2, stride 2) + bias + ReLU, the 1x1 predictor.

    Inside the node the activations and their gradients exist as operand PLANES only (layers/conv.py: PlaneTensor):
    a contraction's epilogue writes the planes the next contraction reads and no fp32 copy; every ReLU gate between
    two contractions rides in the epilogue of the data-gradient launch that produces the gradient, read from the
    gated activation's hi plane; bias gradients are summed from the gradient planes; the transposed convolution never
    materialises a pixel-shuffled copy (conv_transpose2x2_*).  fp32 tensors: the input, the logits, the input
    gradient — and the upsampled features when the caller asks for them (`want_features`).

    apply(x, want_features, w_1, b_1, ..., w_k, b_k, w_deconv, b_deconv, w_pred, b_pred) -> (logits, features | None)."""

    @staticmethod
    def forward(ctx, x, want_features, fan, *params):
        ctx.segment = K.SEGMENT
        ctx.fan = fan   # (layers/grad_fan.py: mask_tower_fused claimed x's fan view, if it is one)
        k = (len(params) - 4) // 2
        hs = [K.PlaneTensor.of(x)]
        for j in range(k):
            hs.append(K.planes_forward(hs[-1], params[2 * j], 1, 1, 1, params[2 * j + 1], True))
        wd, bd, wp, bp = params[2 * k:]
        if want_features:
            u, up = K.planes_conv_transpose2x2_forward(hs[-1], wd, bd, True, fp32=True)
        else:
            u, up = None, K.planes_conv_transpose2x2_forward(hs[-1], wd, bd, True)
        logits = K.planes_forward(up, wp, 1, 0, 1, bp, False, fp32=True)
        ctx.k, ctx.hs, ctx.up, ctx.xshape = k, hs, up, tuple(x.shape)
        ctx.bp = bp                  # (the predictor's bias parameter: its gradient may be delivered from the side stream)
        ctx.u = u                    # (fp32 features, when asked for: the gate of a gradient that arrives through them alone)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*params[0:2 * k:2], wd, wp)
        return logits, u

    @staticmethod
    @once_differentiable
    def backward(ctx, dlogits, du):
        from .elementwise import channel_sum

        K.set_segment(ctx.segment)

        k, hs, up = ctx.k, ctx.hs, ctx.up
        saved = ctx.saved_tensors
        ws, wd, wp = saved[:k], saved[k], saved[k + 1]
        need = ctx.needs_input_grad[3:]                 # per parameter
        grads = [None] * (2 * k + 4)
        if dlogits is None and du is None:
            return (None, None, None) + tuple(grads)
        sums, deconv_summed = K.ColsumBatch(), False
        # ---- predictor; its data gradient lands gated by the upsampler's ReLU, as planes
        if dlogits is not None:
            dl = dlogits.contiguous(memory_format=CL)
            gl = K.PlaneTensor.of(dl, grad=True)
            dbp = torch.empty(wp.shape[0], dtype=torch.float32, device=dl.device) if need[2 * k + 3] else None
            if need[2 * k + 2]:     # (the predictor's bias gradient rides in its weight-gradient contraction)
                def predictor_gradients(gl=gl):
                    db_ = torch.empty(wp.shape[0], dtype=torch.float32, device=dl.device) if dbp is not None else None
                    return [_same_strides(K.planes_backward_weight(gl, up, wp, bias_out=db_), wp), db_]
                # (beside the rest of the backward on the weight-gradient side stream: nothing waits for them)
                if not K.side_weight_gradients([wp, ctx.bp if dbp is not None else None], predictor_gradients,
                                               (gl.buf, up.buf)):
                    grads[2 * k + 2] = _same_strides(K.planes_backward_weight(gl, up, wp, bias_out=dbp), wp)
                    grads[2 * k + 3] = dbp
            elif dbp is not None:
                grads[2 * k + 3] = channel_sum(dl)
            # (a bias gradient is the column sum of its layer's output gradient: taken in the epilogue of the launch that
            # WRITES that gradient — bias_out of the planes_* data gradients — where the shape allows, otherwise from
            # the gradient planes in one batched pass at the end)
            if BIAS_COLSUM and need[2 * k + 1] and du is None:
                gu, deconv_summed = K.planes_backward_data(gl, wp, up.shape, 1, 0, 1, gate=up,
                                                           bias_out=sums.slot(2 * k + 1))
            else:
                gu = K.planes_backward_data(gl, wp, up.shape, 1, 0, 1, gate=up, accumulate=du)
        else:   # a gradient into the upsampled features only (the logits unused): the plain ReLU gate, with planes
            gu = K.PlaneTensor.of(relu_backward(du.contiguous(memory_format=CL), ctx.u, emit_planes=True), grad=True)
        # ---- transposed convolution
        bias_of = []                                   # (slot in grads, gradient planes) still to be summed
        x_device = gu.buf.device
        if need[2 * k] and not K.side_weight_gradients(
                [wd], lambda gu=gu: [K.planes_conv_transpose2x2_backward_weight(gu, hs[k], wd)], (gu.buf, hs[k].buf)):
            grads[2 * k] = K.planes_conv_transpose2x2_backward_weight(gu, hs[k], wd)
        if need[2 * k + 1] and not deconv_summed:
            bias_of.append((2 * k + 1, gu))

        def wants_sum(j):    # layer j's bias gradient is not produced inside its weight-gradient contraction
            return BIAS_COLSUM and j >= 0 and need[2 * j + 1] and not (need[2 * j] and K.wgrad_bias_fits(hs[j].shape, ws[j].shape, 1, 1, 1))

        summed = False                                 # the column sum of the current g has been delivered
        if wants_sum(k - 1):
            g, summed = K.planes_conv_transpose2x2_backward_data(gu, wd, gate=hs[k], bias_out=sums.slot(2 * (k - 1) + 1))
        else:
            g = K.planes_conv_transpose2x2_backward_data(gu, wd, gate=hs[k] if k > 0 else None)
        # ---- the 3x3 tower, last layer first
        dx = None
        for j in range(k - 1, -1, -1):
            w = ws[j]
            inside = need[2 * j] and need[2 * j + 1] and K.wgrad_bias_fits(hs[j].shape, w.shape, 1, 1, 1)
            if need[2 * j]:
                db = torch.empty(w.shape[0], dtype=torch.float32, device=x_device) if inside else None
                if db is None:   # queued: the tower's same-shape layers (both heads) go out as one grouped launch
                    grads[2 * j] = K.planes_backward_weight_deferred(g, hs[j], w, 1, 1, 1)
                else:
                    grads[2 * j] = K.planes_backward_weight(g, hs[j], w, 1, 1, 1, bias_out=db)
                    grads[2 * j + 1] = db
            if need[2 * j + 1] and not inside and not summed:
                bias_of.append((2 * j + 1, g))
            summed = False
            if j > 0:
                if wants_sum(j - 1):
                    g, summed = K.planes_backward_data(g, w, hs[j].shape, 1, 1, 1, gate=hs[j],
                                                       bias_out=sums.slot(2 * (j - 1) + 1))
                else:
                    g = K.planes_backward_data(g, w, hs[j].shape, 1, 1, 1, gate=hs[j])
            elif ctx.needs_input_grad[0]:
                sink = grad_fan.target(ctx.fan, ctx.xshape, g.buf.device)     # (the other head's map: add into it)
                dx = K.planes_backward_data(g, w, ctx.xshape, 1, 1, 1, fp32=True, into=sink)
                if sink is not None or grad_fan.offer(ctx.fan, dx):
                    dx = None
        if k == 0 and ctx.needs_input_grad[0]:
            raise NotImplementedError("mask tower without 3x3 layers")   # (mask_tower_ok requires k >= 1)
        for slot, db in sums.finish().items():      # the epilogues' partial sums, all layers in one launch
            grads[slot] = db
        by_width = {}
        for slot, gp in bias_of:
            by_width.setdefault(gp.shape[1], []).append((slot, gp))
        for group in by_width.values():
            for i0 in range(0, len(group), 8):
                part = group[i0:i0 + 8]
                for (slot, _), db in zip(part, K.planes_channel_sum_multi([gp for _, gp in part])):
                    grads[slot] = db
        return (dx, None, None) + tuple(grads)


def mask_tower_ok(x, convs, deconv, predictor):
    """Can the fused node take these layers?  (Plane arithmetic, 3x3/s1/p1 biased convolutions, the native
    transposed convolution, a predictor whose width the 16-byte epilogue can write.)"""
    if not (ENABLED and MASK_TOWER and K.MATH != "f32" and x.is_cuda and x.shape[0] > 0 and x.dtype == torch.float32):
        return False
    if not convs:
        return False
    for c in convs:
        if c.kernel_size != (3, 3) or c.stride != (1, 1) or c.padding != (1, 1) or c.dilation != (1, 1) or \
                c.bias is None or c.norm is not None or c.in_channels % 32 or c.out_channels % 32:
            return False
    if deconv.bias is None or not K.conv_transpose2x2_ok(x, deconv.weight):
        return False
    return predictor.kernel_size == (1, 1) and predictor.out_channels % 8 == 0 and predictor.bias is not None and \
        predictor.norm is None and predictor.activation is None


def mask_tower_fused(x, convs, deconv, predictor, want_features=True):
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    return _MaskTowerFn.apply(x, bool(want_features), grad_fan.claim(x), *params, deconv.weight, deconv.bias,
                              predictor.weight, predictor.bias)


class _FcStackFn(Function):
    """The box head's fully connected stack (DiscriminativeAdaptionNeck, projects/WSL/wsl/modeling/roi_heads/
    box_head.py; its input rescale: roi_heads_jtsm.py:607-633) as ONE autograd node:
        x' = x * row_scale[r]            folded into the plane split of x (the product is never stored)
        h_j = dropout(relu(fc_j(h_{j-1})))   one MFMA GEMM (bias + ReLU in its epilogue) + one pass that applies the
                                             counter-based dropout mask in place and emits the planes for fc_{j+1}
    Backward: the ReLU-and-dropout gate of every layer is ONE pass over dy (mask read off the stored output: positive
    exactly where the unit was kept and active), and the input gradient leaves fc_1's data-gradient launch already
    multiplied by row_scale (epilogue).  No torch multiply, dropout or masked-scale kernels.

    An optional TAIL — one more linear layer on h_k without ReLU or dropout, e.g. every predictor that reads the box
    features as one concatenated weight — rides in the same node, so that its data gradient can leave its launch
    already gated, rescaled and as planes with fc_k's bias gradient as its column sums (otherwise h_k's gradient
    arrives from outside as an fp32 tensor and takes a gate pass and a pass for the sum).

    apply(x (R, K) dense, row_scale (R,) | None, p, seeds, tail_w (T, out_k) | None, tail_b (T,) | None,
          w_1, b_1, ..., w_k, b_k) -> (h_k (R, out_k), tail output (R, T) | None)."""

    @staticmethod
    def forward(ctx, x, row_scale, p, seeds, tail_w, tail_b, *params):
        from .elementwise import dropout_split_, split_rowscale

        k = len(params) // 2
        r, kin = x.shape
        if row_scale is not None:
            hp = K.PlaneTensor(split_rowscale(x, row_scale), (r, kin, 1, 1))
        else:
            hp = K.PlaneTensor.of(x.view(r, kin, 1, 1))
        hs, ys = [hp], []
        for j in range(k):
            w, b = params[2 * j], params[2 * j + 1]
            w4 = w.view(w.shape[0], w.shape[1], 1, 1)
            if p > 0:
                y = K.planes_forward(hs[-1], w4, 1, 0, 1, b, True, fp32=True)
                yp = K.PlaneTensor(dropout_split_(y, p, seeds[j]), y.shape)
            else:
                y, yp = K.planes_forward(hs[-1], w4, 1, 0, 1, b, True, fp32="both")
            hs.append(yp)
            ys.append(y)
        out = ys[-1].view(r, -1)
        K.planes_put(out, hs[-1].buf)              # the predictor GEMM behind the stack finds its operand planes
        tail_out = None
        if tail_w is not None:
            t4 = tail_w.view(tail_w.shape[0], tail_w.shape[1], 1, 1)
            tail_out = K.planes_forward(hs[-1], t4, 1, 0, 1, tail_b, False, fp32=True).view(r, -1)
        ctx.k, ctx.hs, ctx.inv_keep = k, (hs if tail_w is not None else hs[:-1]), 1.0 / (1.0 - p)
        ctx.row_scale = row_scale
        ctx.has_tail = tail_w is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*ys, *params[0::2], *([tail_w] if tail_w is not None else []))
        return out, tail_out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dtail=None):
        from .elementwise import channel_sum, relu_backward_scaled

        K.set_segment("heads")

        k, hs = ctx.k, ctx.hs
        ys, ws = ctx.saved_tensors[:k], ctx.saved_tensors[k:2 * k]
        need = ctx.needs_input_grad[2:]            # (tail_w, tail_b, w_1, b_1, ...: need[4 + 2 j] is w_j's as before)
        grads = [None] * (2 * k)
        g_tw = g_tb = None
        if dout is None and dtail is None:
            return (None,) * (6 + 2 * k)
        dev = (dout if dout is not None else dtail).device
        dy = dout.contiguous().view(ys[-1].shape) if dout is not None else None
        dx = None
        sums = K.ColsumBatch()
        gp = summed = keep = None     # (the next layer's gated gradient planes, when its data gradient produced them)
        if ctx.has_tail and dtail is not None:
            tw = ctx.saved_tensors[2 * k]
            t4 = tw.view(tw.shape[0], tw.shape[1], 1, 1)
            dl = dtail.contiguous().view(dtail.shape[0], dtail.shape[1], 1, 1)
            gl = K.PlaneTensor.of(dl, grad=True)
            if need[2]:
                g_tb = torch.empty(tw.shape[0], dtype=torch.float32, device=dev) if need[3] else None
                g_tw = K.planes_backward_weight(gl, hs[k], t4, 1, 0, 1, bias_out=g_tb).view(tw.shape)
            elif need[3]:
                g_tb = channel_sum(dl.view(dl.shape[0], -1))
            if dy is None and BIAS_COLSUM and need[5 + 2 * (k - 1)]:
                # h_k is read by the tail alone: its gradient leaves the tail's data gradient gated, rescaled, as planes,
                # with fc_k's bias gradient as the column sums
                keep = torch.full((hs[k].shape[0],), ctx.inv_keep, dtype=torch.float32, device=dev)
                gp, summed = K.planes_backward_data(gl, t4, hs[k].shape, 1, 0, 1, gate=hs[k], row_scale=keep,
                                                    bias_out=sums.slot(2 * (k - 1) + 1))
            else:
                dt = K.planes_backward_data(gl, t4, hs[k].shape, 1, 0, 1, fp32=True)
                dy = dt if dy is None else dy + dt
        for j in range(k - 1, -1, -1):
            w = ws[j]
            w4 = w.view(w.shape[0], w.shape[1], 1, 1)
            if gp is None:
                g, gbuf = relu_backward_scaled(dy, ys[j], ctx.inv_keep)
                gp, summed = K.PlaneTensor(gbuf, ys[j].shape), False
            if need[4 + 2 * j]:
                db = torch.empty(w.shape[0], dtype=torch.float32, device=dev) if (need[5 + 2 * j] and not summed) else None
                # (fc1's 12544 x 2048 weight gradient is 0.33 ms nothing waits for: beside the rest of the backward on the
                # weight-gradient side stream when its bias gradient comes from a data-gradient epilogue anyway)
                if db is None and K.side_weight_gradients(
                        [w], lambda gp=gp, hj=hs[j], w4=w4: [K.planes_backward_weight(gp, hj, w4, 1, 0, 1)],
                        (gp.buf, hs[j].buf)):
                    grads[2 * j] = None
                else:
                    grads[2 * j] = K.planes_backward_weight(gp, hs[j], w4, 1, 0, 1, bias_out=db).view(w.shape)
                if not summed:
                    grads[2 * j + 1] = db
            elif need[5 + 2 * j] and not summed:
                grads[2 * j + 1] = K.planes_channel_sum(gp)
            if j > 0:
                if BIAS_COLSUM and need[5 + 2 * (j - 1)] and ctx.inv_keep > 0:
                    # layer j - 1's gated gradient straight from this data gradient's epilogue: the ReLU-and-dropout gate
                    # read off the planes of its (dropped-out) output, the 1 / (1 - p) as a row factor, the bias
                    # gradient as the column sums — no fp32 copy, no gate pass, no pass for the sum
                    if keep is None:
                        keep = torch.full((hs[j].shape[0],), ctx.inv_keep, dtype=torch.float32, device=dev)
                    gp, summed = K.planes_backward_data(gp, w4, hs[j].shape, 1, 0, 1, gate=hs[j], row_scale=keep,
                                                        bias_out=sums.slot(2 * (j - 1) + 1))
                else:
                    dy, gp = K.planes_backward_data(gp, w4, hs[j].shape, 1, 0, 1, fp32=True), None
            elif ctx.needs_input_grad[0]:
                dx = K.planes_backward_data(gp, w4, hs[0].shape, 1, 0, 1, fp32=True, row_scale=ctx.row_scale)
                dx = dx.view(dx.shape[0], -1)
        for slot, db in sums.finish().items():
            grads[slot] = db
        return (dx, None, None, None, g_tw, g_tb) + tuple(grads)


def fc_stack_ok(x2d, fcs):
    """Plane arithmetic, dense fp32 rows, every width a multiple of 32 (whole 16-byte chunks and K stages)."""
    if not (ENABLED and K.MATH != "f32" and x2d.is_cuda and x2d.dtype == torch.float32 and x2d.dim() == 2 and
            x2d.is_contiguous() and x2d.shape[0] > 0 and x2d.shape[1] % 32 == 0 and len(fcs) > 0):
        return False
    return all(fc.bias is not None and fc.out_features % 32 == 0 and fc.in_features % 32 == 0 for fc in fcs)


def fc_stack_fused(x2d, fcs, row_scale=None, p=0.0, tail=None):
    """tail = (weights, biases) of linear layers that read the stack's output (every predictor of the box head): run
    as ONE more GEMM inside the node (see _FcStackFn); then returns (features, [one output per tail layer])."""
    seeds = tuple(int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in fcs) if p > 0 else None
    params = []
    for fc in fcs:
        params += [fc.weight, fc.bias]
    if tail is None:
        return _FcStackFn.apply(x2d, row_scale, float(p), seeds, None, None, *params)[0]
    weights, biases = list(tail[0]), list(tail[1])
    sizes = [w.shape[0] for w in weights]
    pad = (-sum(sizes)) % 4                      # (16-byte rows: as layers/conv.py linear_fused_split pads)
    if pad:
        weights.append(weights[0].new_zeros((pad, weights[0].shape[1])))
        biases.append(biases[0].new_zeros(pad))
    out, y = _FcStackFn.apply(x2d, row_scale, float(p), seeds, torch.cat(weights), torch.cat(biases), *params)
    outs = K._ColumnSplit.apply(y, *(sizes + ([pad] if pad else [])))
    return out, list(outs[:len(sizes)])
