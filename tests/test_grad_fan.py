"""layers/grad_fan.py: a tensor's consumers add into ONE gradient map instead of autograd adding their maps.
CPU part: the protocol with stand-in consumers (ordinary tensor ops as the ones that do not take part, a small
autograd node as the one that does).  GPU part: the real consumers (both poolers and a convolution) on shared
feature levels."""
import pytest
import torch
from torch.autograd import Function

from jtsm_amd.layers import grad_fan

CL = torch.channels_last


class _Scaled(Function):
    """y = k * x, taking part in the fan protocol (in-place accumulation with tensor ops)."""

    @staticmethod
    def forward(ctx, x, k, rec):
        ctx.k, ctx.rec = k, rec
        return x * k

    @staticmethod
    def backward(ctx, g):
        sink = grad_fan.target(ctx.rec, g.shape, g.device)
        if sink is not None:
            sink.add_(g * ctx.k)
            return None, None, None
        dx = (g * ctx.k).contiguous(memory_format=CL)
        return (None if grad_fan.offer(ctx.rec, dx) else dx), None, None


def _scaled(x, k):
    return _Scaled.apply(x, k, grad_fan.claim(x))


def test_fan_sums_like_autograd_whoever_takes_part():
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 8, 5, 7, generator=g).contiguous(memory_format=CL)
    up = torch.randn(2, 8, 5, 7, generator=g)

    def run(mode):
        x = x0.clone().requires_grad_()
        h = x * 1.5                                  # (the fanned tensor is not a leaf, as in the model)
        if mode == "plain":
            a = b = c = h
        else:
            a, b, c = grad_fan.fan_out(h, 3)
        if mode == "all":                            # every consumer takes part
            y = _scaled(a, 2.0) + _scaled(b, 3.0) + _scaled(c, -1.0)
        elif mode == "mixed":                        # one ordinary consumer among them
            y = _scaled(a, 2.0) + b * 3.0 + _scaled(c, -1.0)
        else:
            y = a * 2.0 + b * 3.0 + c * -1.0
        (y * up).sum().backward()
        return x.grad

    want = run("plain")
    grad_fan.STATS.update(nodes=0, slots_filled=0)
    got = run("all")
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert grad_fan.STATS == {"nodes": 1, "slots_filled": 0}          # one map, no additions by autograd
    grad_fan.STATS.update(nodes=0, slots_filled=0)
    assert torch.allclose(run("mixed"), want, rtol=1e-6, atol=1e-6)
    assert grad_fan.STATS == {"nodes": 1, "slots_filled": 1}
    assert torch.allclose(run("none"), want, rtol=1e-6, atol=1e-6)


def test_a_view_is_claimed_once_and_a_second_backward_starts_clean():
    x = torch.randn(1, 4, 3, 3).contiguous(memory_format=CL).requires_grad_()
    h = x * 1.0
    a, b = grad_fan.fan_out(h, 2)
    assert grad_fan.claim(a) is not None and grad_fan.claim(a) is None          # one claim per view
    a2, b2 = grad_fan.fan_out(h, 2)
    y = _scaled(a2, 2.0) + _scaled(b2, 5.0) + _scaled(b2, 7.0)                 # b2 read twice: the second does not take part
    y.sum().backward(retain_graph=True)
    first = x.grad.clone()
    assert torch.allclose(first, torch.full_like(first, 14.0))
    x.grad = None
    y.sum().backward()
    assert torch.equal(x.grad, first)
    # no gradient wanted / switched off: the tensor itself
    with torch.no_grad():
        assert grad_fan.fan_out(h, 3)[1] is h
    assert grad_fan.fan_out(torch.zeros(2), 3)[0].requires_grad is False


@pytest.mark.gpu
def test_poolers_and_convolution_add_into_one_map_per_level():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from jtsm_amd.layers.wrappers import Conv2d
    from jtsm_amd.modeling.poolers import ROIPooler
    from jtsm_amd.structures import Boxes

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    B, Cc = 2, 256
    scales = (1 / 4, 1 / 8, 1 / 16, 1 / 32)
    feats0 = [torch.randn(B, Cc, 256 // (2 ** i), 256 // (2 ** i), generator=g).to(dev).contiguous(memory_format=CL)
              for i in range(4)]
    boxes = []
    for _ in range(B):
        xy = torch.rand(400, 2, generator=g) * 800
        wh = torch.rand(400, 2, generator=g) ** 2 * 500 + 4
        boxes.append(Boxes(torch.cat([xy, (xy + wh).clamp(max=1023)], 1).to(dev)))
    sp = (torch.arange(1024, device=dev)[:, None] // 32 * 32 + torch.arange(1024, device=dev)[None, :] // 32)
    sp = sp.to(torch.int32)[None].repeat(B, 1, 1)
    oh = [(torch.rand(400, 1024, generator=g) < 0.3).to(torch.int32).to(dev) for _ in range(B)]
    align = ROIPooler(14, scales, 0, "ROIAlignV2")
    moi = ROIPooler(7, scales, 0, "MOIPool")
    convs = [Conv2d(Cc, 128, kernel_size=3, padding=1).to(dev) for _ in range(4)]
    up = [torch.randn(800, Cc, 14, 14, generator=g).to(dev), torch.randn(800, Cc, 7, 7, generator=g).to(dev)]

    def run(fan):
        xs = [f.clone().requires_grad_() for f in feats0]
        hs = [x * 1.0 for x in xs]
        views = [grad_fan.fan_out(h, 3) if fan else (h, h, h) for h in hs]
        y_c = sum((convs[i](views[i][2]) ** 2).mean() for i in range(4))
        y_a = (align([v[1] for v in views], boxes) * up[0]).sum()
        y_m = (moi([v[0] for v in views], boxes, oh_labels_list=oh, superpixels=sp)[0] * up[1]).sum()
        for c in convs:
            c.zero_grad()
        (y_c + 1e-3 * y_a + 1e-3 * y_m).backward()
        return [x.grad for x in xs]

    want = run(False)
    grad_fan.STATS.update(nodes=0, slots_filled=0)
    got = run(True)
    assert grad_fan.STATS == {"nodes": 4, "slots_filled": 0}, grad_fan.STATS     # every level: one map, nothing added
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max())), (a - b).abs().max().item()
    again = run(True)
    for a, b in zip(got, again):
        assert torch.equal(a, b)                                                  # reproducible bit for bit
