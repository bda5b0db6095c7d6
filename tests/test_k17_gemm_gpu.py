"""K17 (csrc/gemm.hip): the 16-bit MFMA GEMM family against f64 torch expressions of the same products on the
same (already rounded) inputs.  Tolerances: the kernel accumulates in f32 and rounds once to the output type, so
the 16-bit outputs are compared at one ulp of that type (bf16 2^-8, fp16 2^-11 relative, + a small absolute term
for cancellation) and the f32 outputs / atomically accumulated weight gradients at 2e-5 of the operand scale."""
import math

import pytest
import torch
from mask_bev_amd import switches

gpu = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


def _rand(shape, dt, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt).to(_dev())


def _tol(dt):
    return {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 2e-5}[dt]


def _close(out, ref, dt, scale):
    """|out - ref| <= one rounding of the output type (relative) + the f32 accumulation error (absolute, in units
    of the operand scale)."""
    err = (out.double() - ref).abs()
    lim = _tol(dt) * ref.abs() + 3e-5 * scale
    bad = err > lim
    assert not bad.any(), f'max err {err.max().item():.3e}; {int(bad.sum())} of {bad.numel()} elements out of tolerance'


SHAPES_NT = [  # (M, N, K)
    (256, 128, 64), (128, 192, 192), (1000, 576, 192), (131, 200, 72), (4096, 768, 3072), (777, 96, 256),
    (65536 // 8, 192, 2048), (33, 8, 8)]


@gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('m,n,k', SHAPES_NT)
def test_nt_plain_and_bias(dt, m, n, k):
    from mask_bev_amd import ops
    x, w = _rand((m, k), dt, 1), _rand((n, k), dt, 2, 1 / math.sqrt(k))
    b = _rand((n,), torch.float32, 3)
    ref = x.double() @ w.double().t()
    out = ops.gemm16_nt(x, w)
    assert out.dtype == dt and out.shape == (m, n)
    _close(out, ref, dt, 1.0)
    out32 = ops.gemm16_nt(x, w, b, out_dtype=torch.float32)
    _close(out32, ref + b.double(), torch.float32, 4.0)


@gpu
@pytest.mark.parametrize('act', ['relu', 'gelu'])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_nt_activation_epilogue(dt, act):
    from mask_bev_amd import ops
    m, n, k = 1500, 320, 192
    x, w = _rand((m, k), dt, 4), _rand((n, k), dt, 5, 1 / math.sqrt(k))
    b = _rand((n,), torch.float32, 6)
    out, pre = ops.gemm16_nt(x, w, b, act=act, want_pre=True)
    ref_pre = x.double() @ w.double().t() + b.double()
    _close(pre, ref_pre, dt, 2.0)
    # the activation is applied to the stored (rounded) pre-activation, as the unfused chain would
    z = pre.double()
    ref = torch.relu(z) if act == 'relu' else torch.nn.functional.gelu(z)
    _close(out, ref, dt, 2.0)
    out32 = ops.gemm16_nt(x, w, b, act=act, out_dtype=torch.float32)
    ref32 = torch.relu(ref_pre) if act == 'relu' else torch.nn.functional.gelu(ref_pre)
    assert (out32.double() - ref32).abs().max() < 2e-5 * 8


@gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('m,n,k', [(256, 128, 128), (1000, 576, 192), (131, 72, 200), (4096, 3072, 768),
                                   (8192, 192, 768), (50, 8, 8)])
def test_nn_data_gradient(dt, m, n, k):
    from mask_bev_amd import ops
    g, w = _rand((m, n), dt, 7), _rand((n, k), dt, 8, 1 / math.sqrt(n))
    ref = g.double() @ w.double()
    _close(ops.gemm16_nn(g, w), ref, dt, 1.0)
    _close(ops.gemm16_nn(g, w, out_dtype=torch.float32), ref, torch.float32, 4.0)


@gpu
@pytest.mark.parametrize('act', ['relu', 'gelu'])
def test_nn_activation_backward_and_colsum(act):
    from mask_bev_amd import ops
    dt = torch.bfloat16
    m, n, k = 2100, 192, 768
    g, w = _rand((m, n), dt, 9), _rand((n, k), dt, 10, 1 / math.sqrt(n))
    aux = _rand((m, k), dt, 11)
    cs = torch.zeros(k, dtype=torch.float32, device=_dev())
    out = ops.gemm16_nn(g, w, act=act, aux=aux, colsum=cs)
    z = aux.double()
    if act == 'relu':
        d = (z > 0).double()
    else:
        d = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    ref = (g.double() @ w.double()) * d
    _close(out, ref, dt, 2.0)
    ref_cs = out.double().sum(0)
    assert (cs.double() - ref_cs).abs().max() < 1e-3 * max(1.0, ref_cs.abs().max().item())


@gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('m,n,k,splits', [(256, 128, 128, 0), (1000, 576, 192, 0), (4099, 200, 72, 0),
                                          (16384, 384, 1536, 0), (70, 768, 192, 1), (65536, 192, 192, 0),
                                          (5000, 96, 256, 7)])
def test_tn_weight_gradient_accumulates(dt, m, n, k, splits):
    from mask_bev_amd import ops
    g, x = _rand((m, n), dt, 12), _rand((m, k), dt, 13)
    acc0 = _rand((n, k), torch.float32, 14)
    acc = acc0.clone()
    ops.gemm16_tn_acc(acc, g, x, splits)
    ref = acc0.double() + g.double().t() @ x.double()
    err = (acc.double() - ref).abs().max().item()
    assert err < 3e-5 * math.sqrt(m) * 4, err


@gpu
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_tn_group_equals_per_layer_products(dt):
    """mbv_gemm16_tn_group: many weight gradients in one launch per 48 — single-part entries (owner adds in place),
    multi-part entries (stored parts + k_add_parts_group), ragged token counts and edges, strided operands, a skipped
    empty entry, more entries than one launch holds.  Against f64 on the same rounded inputs, and bit-reproducible."""
    from mask_bev_amd import ops
    shapes = [(256, 128, 128), (1000, 576, 192), (4099, 200, 72), (16384, 384, 1536), (70, 768, 192), (65536, 192, 192),
              (5000, 96, 256), (21504, 544, 256), (4096, 2304, 768), (1024, 1536, 1536), (8200, 8, 8)]
    shapes = shapes + [(300 + 17 * i, 64 + 8 * (i % 5), 40 + 8 * (i % 3)) for i in range(45)]     # 56 entries: two launches
    items, refs = [], []
    for i, (m, n, k) in enumerate(shapes):
        g, x = _rand((m, n + 8), dt, 100 + i)[:, :n], _rand((m, k), dt, 200 + i)        # g with a row stride
        acc0 = _rand((n, k), torch.float32, 300 + i)
        items.append((g, x, acc0.clone()))
        refs.append((acc0.double() + g.double().t() @ x.double(), m))
    ops.gemm16_tn_group(items)
    for (g, x, acc), (ref, m) in zip(items, refs):
        err = (acc.double() - ref).abs().max().item()
        assert err < 3e-5 * math.sqrt(m) * 4, (tuple(g.shape), tuple(x.shape), err)
    again = [(g, x, _rand(tuple(a.shape), torch.float32, 300 + i)) for i, (g, x, a) in enumerate(items)]
    ops.gemm16_tn_group(again)
    for (_, _, a), (_, _, b) in zip(items, again):
        assert torch.equal(a, b)
    with pytest.raises(ops.MaskBevHipError):
        ops.gemm16_tn_group([(items[0][0], items[0][1], items[0][2].t())])


@gpu
def test_tn_group_with_a_repeated_destination():
    """A weight used twice in one backward pass (tied weights, one Linear applied twice) queues two products on the
    SAME dW.  Inside one grouped launch a destination is updated without atomics, so the launcher must separate them
    (ops.launch_tn_group -> successive launches): both contributions arrive, single-part and multi-part entries alike."""
    from mask_bev_amd import ops
    dt = torch.bfloat16
    items, acc0, refs = [], {}, {}
    for name, (n, k) in {'small': (128, 128), 'deep': (192, 192)}.items():
        acc0[name] = _rand((n, k), torch.float32, 7)
        refs[name] = acc0[name].double().clone()
    accs = {name: a.clone() for name, a in acc0.items()}
    for i, (name, m) in enumerate([('small', 400), ('deep', 20000), ('small', 700), ('deep', 9000), ('small', 400)]):
        n, k = accs[name].shape
        g, x = _rand((m, n), dt, 10 + i), _rand((m, k), dt, 20 + i)
        items.append((g, x, accs[name]))
        refs[name] += g.double().t() @ x.double()
    waves = ops._distinct_destination_waves(items)
    assert [len(w) for w in waves] == [2, 2, 1]
    ops.launch_tn_group(items)
    for name in accs:
        err = (accs[name].double() - refs[name]).abs().max().item()
        assert err < 3e-5 * math.sqrt(30000) * 4, (name, err)


@gpu
def test_linear_applied_twice_in_one_backward_pass():
    """The same through autograd: one arena Linear applied twice to 16-bit inputs; its weight gradient (two deferred
    K17 products on one dW) against the f32 expression on the same rounded operands."""
    from mask_bev_amd import arena as A, layers, ops
    torch.manual_seed(0)
    lin = layers.Linear(192, 192).to(_dev())
    A.ParameterArena([('m', lin)], torch.bfloat16)
    assert getattr(lin.weight, '_mbv_arena', False)
    x1 = _rand((8192, 192), torch.bfloat16, 1).requires_grad_()
    x2 = _rand((8192, 192), torch.bfloat16, 2).requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = lin(x1).float().square().sum() * 1e-3 + lin(x2).float().sum()
    y.backward()
    ops.flush_deferred_grads()
    w = lin.weight.detach().to(torch.bfloat16).double()
    b = lin.bias.detach().double()
    y1 = x1.detach().double() @ w.t() + b
    g1 = (2e-3 * y1).to(torch.bfloat16).double()            # the gradient the 16-bit layer sees
    g2 = torch.ones_like(y1)
    ref = g1.t() @ x1.detach().double() + g2.t() @ x2.detach().double()
    got = lin.weight.grad.double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-2


@gpu
def test_tn_store_batched():
    from mask_bev_amd import ops
    dt = torch.bfloat16
    b, m, n, k = 3, 100, 256, 1032
    g, x = _rand((b, m, n), dt, 15), _rand((b, m, k), dt, 16)
    ref = g.double().transpose(1, 2) @ x.double()
    _close(ops.gemm16_tn(g, x), ref, dt, 10.0)
    _close(ops.gemm16_tn(g, x, out_dtype=torch.float32), ref, torch.float32, 40.0)


@gpu
def test_strided_operands_and_unsupported_shapes():
    from mask_bev_amd import ops
    from mask_bev_amd._lib import MaskBevHipError
    dt = torch.bfloat16
    big = _rand((300, 1152), dt, 17)
    x = big[:, 384:768]                       # a column block of a wider matrix (ld 1152)
    w = _rand((256, 384), dt, 18, 0.05)
    _close(ops.gemm16_nt(x, w), x.double() @ w.double().t(), dt, 1.0)
    with pytest.raises(MaskBevHipError):
        ops.gemm16_nt(_rand((16, 12), dt, 1), _rand((8, 12), dt, 2))      # K % 8 != 0
    with pytest.raises(MaskBevHipError):
        ops.gemm16_nt(torch.zeros(16, 16), torch.zeros(8, 16))            # CPU tensors / f32


@gpu
@pytest.mark.parametrize('act', ['gelu', 'relu'])
def test_fused_ffn_matches_unfused_layers(act, monkeypatch):
    """layers.FFN through ops.ffn (K17 epilogues, arena gradients) against the same module on the library path:
    output and every gradient (x, both weights, both biases), bf16 autocast."""
    from mask_bev_amd import layers, ops
    from mask_bev_amd.arena import ParameterArena
    torch.manual_seed(3)
    c, rows = 192, 9000
    x0 = torch.randn(2, rows // 2, c, device=_dev())
    g0 = torch.randn(2, rows // 2, c, device=_dev()).to(torch.bfloat16)

    def run(policy):
        switches.patch(monkeypatch, gemm16=policy)
        torch.manual_seed(5)
        m = layers.FFN(c, 4 * c, act=act).to(_dev())
        arena = ParameterArena([('ffn', m)])
        x = x0.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert ops.ffn_fused_ok(x, m.layers[0][0].weight, m.layers[0][0].bias, m.layers[1].weight,
                                    m.layers[1].bias) == (policy != '0')
            y = m(x, add_identity=False)
        y.backward(g0)
        return y.float(), x.grad.float(), arena.grad.clone()

    y1, gx1, ga1 = run('auto')
    y0, gx0, ga0 = run('0')
    y2, gx2, ga2 = run('all')
    # bf16 tolerance 2e-2 of the largest value; ReLU' is discontinuous, and the two paths round the fc1 bias
    # differently (f32 here, the bf16 shadow in the library path), so ~1e-3 of the hidden units near 0 switch side:
    # its gradients are compared in the L2 norm (1e-2), which isolated switches do not dominate
    for y, gx, ga in ((y1, gx1, ga1), (y2, gx2, ga2)):
        assert (y - y0).abs().max() <= 2e-2 * y0.abs().max()
        if act == 'gelu':
            assert (gx - gx0).abs().max() <= 2e-2 * gx0.abs().max()
            assert (ga - ga0).abs().max() <= 2e-2 * ga0.abs().max()
        assert (gx - gx0).norm() <= 1e-2 * gx0.norm()
        assert (ga - ga0).norm() <= 1e-2 * ga0.norm()


@gpu
def test_ffn_bias_gradient_deferred_rows_equal_immediate_reduction(monkeypatch):
    """mbv_gemm16_nn_parts: inside a backward pass the fused data gradient leaves its per-wave-row column sums as rows of
    a tensor that joins the pass's grouped column-sum launch; the fc1 bias gradient equals the immediate reduction's (the
    two add the same f32 partial rows in a different order) and the sum of d(hidden) the library path forms."""
    from mask_bev_amd import layers, ops
    from mask_bev_amd.arena import ParameterArena
    c, rows = 192, 8200            # 8200 rows: a ragged last tile (its dead wave rows must contribute zeros)
    torch.manual_seed(11)
    x0 = torch.randn(rows, c, device=_dev())
    g0 = torch.randn(rows, c, device=_dev()).to(torch.bfloat16)

    def run(defer):
        switches.patch(monkeypatch, nn_colsum_defer=defer)
        torch.manual_seed(5)
        m = layers.FFN(c, 4 * c, act='gelu').to(_dev())
        arena = ParameterArena([('ffn', m)])
        x = x0.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = m(x, add_identity=False)
        y.backward(g0)
        return m.layers[0][0].bias.grad.clone(), arena.grad.clone()

    b_def, a_def = run(True)
    b_now, a_now = run(False)
    assert b_def.abs().max() > 0
    assert (b_def - b_now).abs().max() <= 1e-5 * b_now.abs().max()
    assert (a_def - a_now).abs().max() <= 1e-5 * a_now.abs().max()
    lib = ops._lib.load()
    for m_, k_ in ((8200, 768), (64, 8), (65536, 768), (129, 192)):
        r = lib.mbv_gemm16_nn_part_rows(m_, k_, 1)
        assert (m_ + 63) // 64 <= r and r * k_ * 4 <= lib.mbv_gemm16_nn_workspace_bytes(m_, k_, 1)


@gpu
def test_mask_logit_backward_products():
    """ops.mask_logits_backward: d_embed = dl . F^T (split-K NT, f32 atomics) and d_feature = E^T . dl (batched TN)
    at the deferred-head shape of the bench (rows = 10 outputs x 100 queries, C = 256) on a reduced pixel count."""
    from mask_bev_amd import ops
    dt = torch.bfloat16
    b, r, c, p = 2, 1000, 256, 4096
    dl, e, f = _rand((b, r, p), dt, 21, 0.05), _rand((b, r, c), dt, 22), _rand((b, c, p), dt, 23)
    g_e, g_f = ops.mask_logits_backward(dl, e, f)
    ref_e = dl.double() @ f.double().transpose(1, 2)
    ref_f = e.double().transpose(1, 2) @ dl.double()
    assert g_e.dtype == torch.float32 and g_f.dtype == dt
    assert (g_e.double() - ref_e).abs().max() < 2e-4 * ref_e.abs().max()
    _close(g_f, ref_f, dt, float(ref_f.abs().max()) * 1e-2)


@gpu
@pytest.mark.parametrize('bias', [True, False])
def test_conv1x1_as_batched_gemm_equals_conv2d(bias):
    """layers.conv1x1 (the pixel decoder's 1 x 1 projections as one strided-batched GEMM,
    mask_bev_panoptic_head.py:119-123 / mmdet MSDeformAttnPixelDecoder's ConvModules) against nn.Conv2d itself in fp32:
    output, input gradient, weight and bias gradients."""
    from mask_bev_amd.layers import conv1x1
    device = _dev()
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(96, 64, 1, bias=bias).to(device)
    x = torch.randn(3, 96, 20, 28, device=device, requires_grad=True)
    go = torch.randn(3, 64, 20, 28, device=device)
    y_ref = conv(x)
    y_ref.backward(go)
    want = [y_ref.detach(), x.grad.clone(), conv.weight.grad.clone()] + ([conv.bias.grad.clone()] if bias else [])
    x.grad = None
    conv.zero_grad()
    y = conv1x1(conv, x)
    y.backward(go)
    got = [y.detach(), x.grad, conv.weight.grad] + ([conv.bias.grad] if bias else [])
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)


@gpu
@pytest.mark.parametrize('bias', [True, False])
@pytest.mark.parametrize('autocast', [None, torch.bfloat16])
def test_conv1x1_of_a_channels_last_view(bias, autocast):
    """The backbone hands its stage outputs over as (B, C, H, W) VIEWS of channels-last token maps; layers.conv1x1 reads
    them as the transposed GEMM operand (ops.conv1x1_tokens) and its backward returns a token-major gradient — against
    nn.Conv2d on a contiguous copy: output, input gradient (and that it is channels-last, i.e. no copy is needed to hand
    it back to the backbone), weight and bias gradients.  fp32 exact to 2e-4; under bf16 autocast to one bf16 rounding."""
    from mask_bev_amd.layers import conv1x1
    device = _dev()
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(96, 64, 1, bias=bias).to(device)
    tokens = torch.randn(3, 20, 28, 96, device=device, requires_grad=True)          # (B, H, W, C)
    go = torch.randn(3, 64, 20, 28, device=device)
    xr = tokens.detach().permute(0, 3, 1, 2).contiguous().requires_grad_()
    y_ref = conv(xr)
    y_ref.backward(go)
    want = [y_ref.detach(), xr.grad.permute(0, 2, 3, 1), conv.weight.grad.clone()] + ([conv.bias.grad.clone()] if bias else [])
    conv.zero_grad()
    with torch.autocast('cuda', dtype=autocast or torch.bfloat16, enabled=autocast is not None):
        y = conv1x1(conv, tokens.permute(0, 3, 1, 2))
    y.backward(go.to(y.dtype))
    assert tokens.grad.is_contiguous()
    got = [y.detach().float(), tokens.grad, conv.weight.grad] + ([conv.bias.grad] if bias else [])
    tol = 2e-4 if autocast is None else 2e-2
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + tol * 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('b,c,cout,h,w', [(2, 32, 64, 24, 23), (4, 256, 256, 32, 32)])
def test_conv3x3_on_k17_matches_float64(dtype, b, c, cout, h, w):
    """The 3 x 3 convolution of the 16-bit compute modes as K17 products on zero-bordered channels-last rows
    (mbv_conv3x3_gemm16; the weight gradient as nine grouped TN entries) against the float64 convolution of the same 16-bit
    operands: output and input gradient within the 16-bit storage rounding, the f32 weight gradient within 1e-3 (bf16 dy)."""
    from mask_bev_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(c + h)
    conv = torch.nn.Conv2d(c, cout, 3, padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(conv.weight.to(dtype).float())
    x = torch.randn(b, c, h, w, generator=g).to(dtype).to(dev).requires_grad_()
    gy = (torch.randn(b, cout, h, w, generator=g) * 1e-2).to(dtype).to(dev)
    assert ops.conv3x3_16_ok(x, conv)
    y = ops.conv3x3_16(x, conv.weight)
    assert y.dtype == dtype
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(), conv.weight.detach().double().requires_grad_()
    ref = torch.nn.functional.conv2d(xd, wd, padding=1)
    ref.backward(gy.double())

    def err(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert err(y.detach(), ref.detach()) <= 1.5 * eps
    assert err(x.grad, xd.grad) <= 1.5 * eps
    assert err(conv.weight.grad, wd.grad) <= 1e-3
