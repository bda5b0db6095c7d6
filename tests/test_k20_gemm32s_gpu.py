"""K20 (csrc/gemm_f32s.hip): f32 GEMMs formed on the 16-bit matrix pipe from IEEE-half PAIRS of the scaled operands
(hi.hi + hi.lo + lo.hi, f32 accumulation), against the float64 product of the same f32 operands.

Bar: max-norm error <= 2e-6 of max|ref| — the level `test_skinny_gemm_f32_equals_float64_product` holds the
exact-f32 MFMA kernel to — on every layout (NT forward, NN data gradient, TN weight gradient), ragged edges, operand
magnitudes from 1e-9 to 1e+6 (the per-tensor power-of-two scales from `mbv_f32_absmax_group`), and next to the library's own
f32 GEMM on the same operands (the split product must not be worse than 4x the f32 GEMM's error)."""
import pytest
import torch
from mask_bev_amd import switches

gpu = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(_dev())


def _err(out, ref):
    return float((out.double() - ref).abs().max() / ref.abs().max().clamp(min=1e-300))


@gpu
def test_absmax_words_are_the_bits_of_the_maximum():
    from mask_bev_amd import ops
    a = _rand((1000, 192), 1, 3.0)
    wide = _rand((257, 512), 2, 1e-5)
    b = wide[:, 128:328]                                   # strided rows, offset start (200 columns, 16-byte aligned)
    z = torch.zeros(64, 8, device=_dev())
    big = _rand((65536, 96), 3, 100.0)
    words = ops.f32_absmax([a, b, z, big])
    assert tuple(words.shape) == (4, ops.AMAX_SLOTS)        # a record = 64 words whose maximum is the result
    want = torch.stack([t.abs().max() for t in (a, b, z, big)]).view(torch.int32)
    assert torch.equal(words.max(1).values.cpu(), want.cpu())
    # a second launch into the same words keeps the larger of the two (integer max): the grouped-weights use
    from mask_bev_amd import _lib
    import ctypes
    lib = _lib.load()
    small = (a * 0.5).contiguous()
    PA, LA = ctypes.c_void_p * 1, ctypes.c_int64 * 1
    ops.check(lib.mbv_f32_absmax_group(PA(small.data_ptr()), LA(1000), LA(192), LA(192), PA(words.data_ptr()), 1,
                                       ops._stream()), 'mbv_f32_absmax_group')
    assert int(words[0].max()) == int(want[0])


SHAPES = [  # (M, N, K)
    (256, 128, 64), (128, 192, 192), (1000, 576, 192), (131, 200, 72), (4096, 768, 3072), (777, 96, 256), (33, 8, 8),
    (8192, 192, 2048), (5000, 1024, 40)]


@gpu
@pytest.mark.parametrize('m,n,k', SHAPES)
@pytest.mark.parametrize('sx,sw', [(1.0, 0.05), (3e-7, 40.0), (2e5, 1e-6)])
def test_nt_equals_float64_product(m, n, k, sx, sw):
    from mask_bev_amd import ops
    x, w = _rand((m, k), m + k, sx), _rand((n, k), n + k + 1, sw)
    bias = _rand((n,), 5, sx * sw * k ** 0.5)
    ref = x.double() @ w.double().t() + bias.double()
    out = ops.gemm32s_nt(x, w, bias)
    assert torch.isfinite(out).all()
    e = _err(out, ref)
    lib_e = _err(torch.addmm(bias, x, w.t()), ref)
    assert e <= 2e-6, (e, lib_e)
    assert e <= max(4 * lib_e, 5e-7), (e, lib_e)


@gpu
def test_nt_activation_epilogues_and_strided_operands():
    from mask_bev_amd import ops
    m, n, k = 3000, 768, 192
    xw = _rand((m, 2 * k), 1)
    x = xw[:, k:]                                          # row stride 2k, offset start
    wfull = _rand((n + 16, k), 2, 0.1)
    w = wfull[8:8 + n]                                     # a row block of a packed weight (the `rows=` use of ops.linear)
    bias = _rand((n,), 3)
    pre_ref = x.double() @ w.double().t() + bias.double()
    for act, fn in (('relu', torch.relu), ('gelu', torch.nn.functional.gelu)):
        out, pre = ops.gemm32s_nt(x, w, bias, act=act, want_pre=True)
        assert _err(pre, pre_ref) <= 2e-6
        assert _err(out, fn(pre_ref)) <= 4e-6             # (the erf of the GELU epilogue is Abramowitz-Stegun 7.1.26: 1.5e-7)
    out = ops.gemm32s_nt(x, w, None)
    assert _err(out, pre_ref - bias.double()) <= 2e-6


@gpu
@pytest.mark.parametrize('m,n,k', SHAPES)
@pytest.mark.parametrize('sg,sw', [(1.0, 0.05), (1e-8, 0.3)])
def test_nn_equals_float64_product(m, n, k, sg, sw):
    """data gradient: g (M, N) . w (N, K) — w read with the transposing LDS reads from its hi / lo images"""
    from mask_bev_amd import ops
    g, w = _rand((m, n), m + n, sg), _rand((n, k), n + k + 2, sw)
    ref = g.double() @ w.double()
    out = ops.gemm32s_nn(g, w)
    assert torch.isfinite(out).all()
    e, lib_e = _err(out, ref), _err(g @ w, ref)
    assert e <= 2e-6 and e <= max(4 * lib_e, 5e-7), (e, lib_e)


@gpu
@pytest.mark.parametrize('m,n,k', [(512, 128, 64), (4096, 192, 192), (65536, 576, 192), (16384, 768, 3072), (1000, 200, 72),
                                   (8191, 96, 384), (40000, 8, 8)])
@pytest.mark.parametrize('sg', [1.0, 1e-7])
def test_tn_accumulates_the_weight_gradient(m, n, k, sg):
    """acc (N, K) += g (M, N)^T . x (M, K): one part adds in place, several parts are stored and added by their owner —
    either way onto what acc already held, and bit-reproducibly."""
    from mask_bev_amd import ops
    g, x = _rand((m, n), m + n, sg), _rand((m, k), m + k + 3)
    start = _rand((n, k), 9, sg * m ** 0.5)
    ref = start.double() + g.double().t() @ x.double()
    acc = start.clone()
    ops.gemm32s_tn_acc(acc, g, x)
    assert torch.isfinite(acc).all()
    e, lib_e = _err(acc, ref), _err(torch.addmm(start, g.t(), x), ref)
    assert e <= 2e-6 and e <= max(4 * lib_e, 5e-7), (e, lib_e)
    again = start.clone()
    ops.gemm32s_tn_acc(again, g, x)
    assert torch.equal(acc, again)


@gpu
def test_rows_far_below_the_operand_maximum_keep_their_relative_accuracy_down_to_2_pow_minus_18():
    """Per-TENSOR scales: an element 2^-j of the operand's maximum keeps all 22 bits for j <= 18 (csrc/gemm_f32s.hip).  Token
    rows 1e-5 of the largest row (padded / background tokens next to object tokens) are still right to 1e-6 of THEIR OWN
    scale; rows 1e-9 of it only in absolute terms — stated, not hidden."""
    from mask_bev_amd import ops
    m, n, k = 2048, 192, 192
    x, w = _rand((m, k), 1), _rand((n, k), 2, 0.1)
    x[1::2] *= 1e-5
    x[3::8] *= 1e-4                                        # these rows: 1e-9 of the maximum
    ref = x.double() @ w.double().t()
    out = ops.gemm32s_nt(x, w)
    big, small, tiny = ref[0::2], ref[1::8], ref[3::8]
    assert _err(out[0::2], big) <= 2e-6
    assert _err(out[1::8], small) <= 2e-6                   # 1e-5 of the maximum: full accuracy relative to themselves
    assert float((out[3::8].double() - tiny).abs().max()) <= 2e-6 * float(small.abs().max())
    assert _err(out[3::8], tiny) <= 2e-2                    # still the right numbers, with fewer bits


@gpu
def test_unscaled_mode_and_nonfinite_inputs():
    from mask_bev_amd import ops, _lib
    lib = _lib.load()
    m, n, k = 512, 128, 96
    x, w = _rand((m, k), 1), _rand((n, k), 2)
    out = torch.empty(m, n, device=_dev())
    ops.check(lib.mbv_gemm32s_nt(ops._ptr(x), ops._ptr(w), None, ops._ptr(out), None, m, n, k, k, k, n, None, None, None, 0, 1,
                                 0, 0, 0, ops._stream()), 'mbv_gemm32s_nt')
    assert _err(out, x.double() @ w.double().t()) <= 2e-6  # O(1) operands need no scale
    x[7, 5] = float('inf')
    w[3, 9] = float('nan')
    out = ops.gemm32s_nt(x, w)
    assert not torch.isfinite(out[7]).any() and not torch.isfinite(out[:, 3]).any()
    ok = torch.ones(m, dtype=torch.bool)
    ok[7] = False
    # (an infinite maximum makes the scale 2^-115: the finite rows flush to zero instead of being right — the output of a
    # GEMM with a non-finite operand is only required to SHOW it)
    assert lib.mbv_gemm32s_supported(0, 512, 128, 96) == 1 and lib.mbv_gemm32s_supported(0, 512, 100, 96) == 0


@gpu
@pytest.mark.parametrize('arena', [False, True])
def test_fp32_linear_takes_k20_and_matches_float64(arena):
    """ops.linear in fp32 compute with >= gemm32s_min tokens: forward, data gradient, weight and bias gradients against
    float64 — plain parameters (autograd's .grad) and arena parameters (accumulated in place)."""
    from mask_bev_amd import ops
    from mask_bev_amd.arena import ParameterArena
    torch.manual_seed(3)
    lin = torch.nn.Linear(192, 576).to(_dev())
    x = _rand((4, 1024, 192), 4).requires_grad_()
    gy = _rand((4, 1024, 576), 5, 1e-4)
    if arena:
        ParameterArena([('l', lin)], shadow_dtype=None)
    from mask_bev_amd import ops_gemm as G          # (the Linear resolves gemm32s_nt in its own module, not through the `ops` facade)
    calls = []
    orig = G.gemm32s_nt
    G.gemm32s_nt = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        y = ops.linear(x, lin.weight, lin.bias)
    finally:
        G.gemm32s_nt = orig
    assert calls, 'the f32 Linear did not take K20'
    y.backward(gy)
    ops.flush_deferred_grads()
    xd, wd, bd, gd = x.detach().double(), lin.weight.detach().double(), lin.bias.detach().double(), gy.double()
    assert _err(y.detach(), xd @ wd.t() + bd) <= 2e-6
    assert _err(x.grad, gd @ wd) <= 2e-6
    assert _err(lin.weight.grad, gd.flatten(0, 1).t() @ xd.flatten(0, 1)) <= 2e-6
    assert _err(lin.bias.grad, gd.flatten(0, 1).sum(0)) <= 2e-6
    with switches.override(gemm32s=False):
        calls.clear()
        G.gemm32s_nt = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            y2 = ops.linear(x.detach(), lin.weight, lin.bias)
        finally:
            G.gemm32s_nt = orig
        assert not calls and _err(y2.detach(), y.detach().double()) <= 4e-6


@gpu
@pytest.mark.parametrize('b,c,h,w,e', [(2, 8, 16, 128, 24), (4, 128, 64, 256, 192), (1, 32, 256, 128, 96)])
@pytest.mark.parametrize('arena', [False, True])
def test_patch_projection_gathers_the_nchw_image(b, c, h, w, e, arena):
    """K20's GATHER modes: the 4 x 4 stride-4 Conv2d of mmdet's PatchEmbed on the f32 pseudo-image — forward tokens, image
    gradient (scattered back to NCHW, every element written) and weight / bias gradients against float64 convolution."""
    from mask_bev_amd import ops
    from mask_bev_amd.arena import ParameterArena
    torch.manual_seed(b + c + e)
    conv = torch.nn.Conv2d(c, e, 4, stride=4).to(_dev())
    x = _rand((b, c, h, w), 3, 2.0).requires_grad_()
    gy = _rand((b, h // 4, w // 4, e), 4, 1e-3)
    if arena:
        ParameterArena([('c', conv)], shadow_dtype=None)
    assert ops.patch_embed32_ok(x, conv.weight, conv.bias)
    y = ops.patch_embed32(x, conv.weight, conv.bias)
    y.backward(gy)
    ops.flush_deferred_grads()
    xd = x.detach().double().requires_grad_()
    wd, bd = conv.weight.detach().double().requires_grad_(), conv.bias.detach().double().requires_grad_()
    ref = torch.nn.functional.conv2d(xd, wd, bd, stride=4).permute(0, 2, 3, 1)
    ref.backward(gy.double())
    assert y.shape == ref.shape and _err(y.detach(), ref.detach()) <= 2e-6
    assert _err(x.grad, xd.grad) <= 2e-6
    assert _err(conv.weight.grad, wd.grad) <= 2e-6
    assert _err(conv.bias.grad, bd.grad) <= 2e-6


@gpu
def test_patch_projection_refuses_what_it_cannot_gather():
    from mask_bev_amd import ops
    conv = torch.nn.Conv2d(8, 24, 4, stride=4).to(_dev())
    assert not ops.patch_embed32_ok(_rand((1, 8, 16, 96), 1), conv.weight, conv.bias)          # token rows of 24: not 32-aligned
    assert not ops.patch_embed32_ok(_rand((1, 8, 16, 128), 1).half(), conv.weight, conv.bias)
    conv2 = torch.nn.Conv2d(8, 24, 2, stride=2).to(_dev())
    assert not ops.patch_embed32_ok(_rand((1, 8, 16, 128), 1), conv2.weight, conv2.bias)


@gpu
def test_producers_leave_absmax_records_and_consumers_use_them():
    """The absmax HINTS (ops.amax_hint_*): K20's epilogue and the activation / attention wrappers leave a 64-word
    record whose maximum is (a bound of) max|tensor|; a Linear that finds one skips the absmax pass over its operand — same
    results to the last bit as with the pass (a record bounds the same binade unless it is a looser bound, which only moves
    the scale)."""
    from mask_bev_amd import ops, _lib
    x, w = _rand((4096, 192), 1, 2.0), _rand((576, 192), 2, 0.05)
    out = ops.gemm32s_nt(x, w, None, hint_out=True)
    rec = ops.amax_hint_get(out)
    assert rec is not None and int(rec.max()) == int(out.abs().max().view(torch.int32))
    assert ops.amax_hint_get(out.view(4, 1024, 576)) is rec              # a view of the same elements, while `out` lives
    assert ops.amax_hint_get(out[:2048]) is rec                          # a slice of it: the whole tensor's record bounds it
    assert ops.amax_hint_get(out[:2048].clone()) is None                 # another tensor
    out.add_(1.0)
    assert ops.amax_hint_get(out) is None                                # modified in place: the record no longer describes it
    # a Linear behind a hinted tensor runs no absmax pass over it
    lin = torch.nn.Linear(192, 576).to(_dev())
    lib = _lib.load()
    calls = []
    lib.hook = lambda name, fn, args: (calls.append(name), fn(*args))[1]
    ctx = switches.override(amax_hints=True)
    ctx.__enter__()
    try:
        h = ops.gemm32s_nt(x, w[:192].contiguous(), None, hint_out=True)      # (4096, 192) with a record
        calls.clear()
        y1 = ops.linear(h, lin.weight, lin.bias)
        first = list(calls)
        calls.clear()
        y2 = ops.linear(h, lin.weight, lin.bias)                              # the weight's record is cached until parameters change
        second = list(calls)
    finally:
        lib.hook = None
        ctx.__exit__(None, None, None)
    assert first.count('mbv_f32_absmax_group') == 1 and second.count('mbv_f32_absmax_group') == 0, (first, second)
    with switches.override(amax_hints=False):
        y3 = ops.linear(h, lin.weight, lin.bias)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    ops.note_parameters_changed()
    calls = []
    lib.hook = lambda name, fn, args: (calls.append(name), fn(*args))[1]
    try:
        with switches.override(amax_hints=True):
            ops.linear(h, lin.weight, lin.bias)
    finally:
        lib.hook = None
    assert calls.count('mbv_f32_absmax_group') == 1                      # the weight's record again, after an update


@gpu
def test_grouped_few_row_weight_gradients():
    """mbv_gemm32s_tn_group: the weight gradients of many few-row Linears (the decoder's 400 rows) in one launch, each tile
    added in place by its owner — against float64, on top of what the destinations held, bit-reproducibly; and through the
    end-of-pass flush of a backward pass (ops.flush_deferred_grads takes the f32 few-row products there)."""
    from mask_bev_amd import ops
    shapes = [(400, 256, 256), (400, 2048, 256), (400, 256, 2048), (100, 256, 264), (4000, 256, 256), (37, 8, 8)] * 9
    items, refs = [], []
    for i, (m, n, k) in enumerate(shapes):
        g, x = _rand((m, n), 100 + i, 10.0 ** (-(i % 5))), _rand((m, k), 200 + i)
        acc = _rand((n, k), 300 + i, 0.1)
        refs.append(acc.double() + g.double().t() @ x.double())
        items.append((g, x, acc))
    starts = [a.clone() for _, _, a in items]
    ops.gemm32s_tn_group(items)
    for (g, x, acc), ref, s0 in zip(items, refs, starts):
        e, lib_e = _err(acc, ref), _err(torch.addmm(s0, g.t(), x), ref)
        assert e <= max(2e-6, 2 * lib_e), (tuple(g.shape), e, lib_e)      # (a 4 000-term f32 sum: the library's own error is 1-2e-6)
    again = [(g, x, s0) for (g, x, _), s0 in zip(items, starts)]
    ops.gemm32s_tn_group(again)
    assert all(torch.equal(a, b) for (_, _, a), (_, _, b) in zip(items, again))


@gpu
def test_grouped_token_major_weight_gradients():
    """The same launch with token-major products (their token sums cut into ranges of ~ 4 096, partial tiles added by their
    owner), few-row ones beside them, operands with and without their own absmax records, a ragged last range, strided rows:
    float64, bit-reproducible, and equal to what the per-layer entry point gives up to the order of the range sums."""
    from mask_bev_amd import ops
    shapes = [(65536, 192, 192), (16384, 384, 1536), (20001, 264, 136), (4096, 768, 768), (400, 256, 256), (9000, 8, 2048),
              (4097, 128, 128)]
    items, refs = [], []
    for i, (m, n, k) in enumerate(shapes):
        gfull, x = _rand((m, n + 8), 400 + i, 10.0 ** (-(i % 4))), _rand((m, k), 500 + i)
        g = gfull[:, 4:4 + n] if i % 2 else gfull[:, :n].contiguous()            # (odd entries: strided rows, 16-byte aligned)
        acc = _rand((n, k), 600 + i, 0.1)
        refs.append(acc.double() + g.double().t() @ x.double())
        rec = ops.f32_absmax([g, x]) if i % 3 == 0 else None
        items.append((g, x, acc) if rec is None else (g, x, acc, rec[0:1], rec[1:2]))
    starts = [it[2].clone() for it in items]
    ops.gemm32s_tn_group(items)
    for it, ref, s0 in zip(items, refs, starts):
        g, x, acc = it[:3]
        e, lib_e = _err(acc, ref), _err(torch.addmm(s0, g.t(), x), ref)
        assert e <= max(2e-6, 2 * lib_e), (tuple(g.shape), e, lib_e)
    again = [(it[0], it[1], s0.clone()) + tuple(it[3:]) for it, s0 in zip(items, starts)]
    ops.gemm32s_tn_group(again)
    assert all(torch.equal(a[2], b[2]) for a, b in zip(items, again))
    for it, s0, ref in zip(items[:3], starts, refs):
        one = s0.clone()
        ops.gemm32s_tn_acc(one, it[0], it[1])
        assert _err(one, ref) <= 2e-6 and _err(one, it[2].double()) <= 2e-6


@gpu
@pytest.mark.parametrize('kind', ['gelu', 'relu'])
@pytest.mark.parametrize('rows,c,f', [(4096, 192, 768), (2100, 256, 1024)])
def test_fp32_ffn_on_k20_matches_float64(kind, rows, c, f):
    """ops.ffn32 (K20: fc1 + bias + activation in one launch; dgrad(fc2) x act' + the partial column sums of d b1 in one) against
    the float64 FFN: output, input gradient, the four parameter gradients (accumulated into the arena in place)."""
    from mask_bev_amd import ops
    from mask_bev_amd.arena import ParameterArena
    torch.manual_seed(rows + f)
    fc1, fc2 = torch.nn.Linear(c, f).to(_dev()), torch.nn.Linear(f, c).to(_dev())
    ParameterArena([('ffn', torch.nn.ModuleList([fc1, fc2]))], shadow_dtype=None)
    x = _rand((rows, c), 5).requires_grad_()
    gy = _rand((rows, c), 6, 1e-3)
    assert ops.ffn32_ok(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    y = ops.ffn32(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, kind)
    y.backward(gy)
    ops.flush_deferred_grads()
    act = torch.nn.functional.gelu if kind == 'gelu' else torch.relu
    xd = x.detach().double().requires_grad_()
    ps = [p.detach().double().requires_grad_() for p in (fc1.weight, fc1.bias, fc2.weight, fc2.bias)]
    ref = torch.nn.functional.linear(act(torch.nn.functional.linear(xd, ps[0], ps[1])), ps[2], ps[3])
    ref.backward(gy.double())
    assert _err(y.detach(), ref.detach()) <= 4e-6
    assert _err(x.grad, xd.grad) <= (4e-6 if kind == 'gelu' else 2e-5)      # (ReLU: an input within rounding of the kink flips its gate)
    for p, r in zip((fc1.weight, fc1.bias, fc2.weight, fc2.bias), ps):
        assert _err(p.grad, r.grad) <= (6e-6 if kind == 'gelu' else 3e-5), float(_err(p.grad, r.grad))


@gpu
@pytest.mark.parametrize('b,c,cout,h,w', [(2, 32, 64, 24, 23), (1, 64, 32, 33, 37), (4, 256, 256, 32, 32)])
def test_conv3x3_on_k20_matches_float64(b, c, cout, h, w):
    """The 3 x 3 convolution as K20 products on the zero-bordered channels-last rows (mbv_conv_pad_rows / mbv_conv3x3_gemm32s /
    nine grouped TN entries) against the float64 convolution: output, input gradient, weight gradient — and what MIOpen's f32
    convolution gives beside it."""
    from mask_bev_amd import ops
    conv = torch.nn.Conv2d(c, cout, 3, padding=1, bias=False).to(_dev())
    x = _rand((b, c, h, w), 11).requires_grad_()
    gy = _rand((b, cout, h, w), 12, 1e-3)
    assert ops.conv3x3_32_ok(x, conv)
    y = ops.conv3x3_32(x, conv.weight)
    y.backward(gy)
    gx, gw = x.grad.clone(), conv.weight.grad.clone()
    xd, wd = x.detach().double().requires_grad_(), conv.weight.detach().double().requires_grad_()
    ref = torch.nn.functional.conv2d(xd, wd, padding=1)
    ref.backward(gy.double())
    x.grad = None
    conv.weight.grad = None
    lib_y = conv(x)
    lib_y.backward(gy)
    for name, mine, r, lib_v in (('y', y.detach(), ref.detach(), lib_y.detach()), ('dx', gx, xd.grad, x.grad),
                                 ('dw', gw, wd.grad, conv.weight.grad)):
        e, le = _err(mine, r), _err(lib_v, r)
        assert e <= max(2e-6, 2 * le), (name, e, le)


@gpu
def test_layernorm_output_bound_records():
    """ops.ln_bound: word 0 of the record is the bits of sqrt(C) max|gamma| + max|beta| — never below the true maximum of the
    LayerNorm's output, for any input — refreshed (for every LayerNorm seen so far, in one launch) when the parameters change;
    a K20 Linear that finds it as its input's hint stays within the usual 2e-6 of the float64 product."""
    import math
    from mask_bev_amd import ops, switches
    lns = [torch.nn.LayerNorm(c).to(_dev()) for c in (192, 768, 1536)]
    for i, ln in enumerate(lns):
        with torch.no_grad():
            ln.weight.copy_(_rand((ln.weight.numel(),), 40 + i, 2.0))
            ln.bias.copy_(_rand((ln.bias.numel(),), 50 + i, 0.3))
    recs = [ops.ln_bound(ln.weight, ln.bias) for ln in lns]
    for ln, rec in zip(lns, recs):
        c = ln.weight.numel()
        want = math.sqrt(c) * float(ln.weight.abs().max()) + float(ln.bias.abs().max())
        got = float(rec.max().view(torch.float32))
        assert abs(got - want) <= 1e-5 * want
        x = _rand((4096, c), 60, 1.0)
        x[7] = 0.0
        x[7, 3] = 1e4                                 # one spike per row: xhat reaches sqrt(C - 1)
        y = torch.nn.functional.layer_norm(x, (c,), ln.weight, ln.bias)
        assert float(y.abs().max()) <= got
    assert ops.ln_bound(lns[0].weight, lns[0].bias) is recs[0]          # cached until the parameters change
    with torch.no_grad():
        lns[1].weight.mul_(3.0)
    ops.note_parameters_changed()
    r1 = ops.ln_bound(lns[1].weight, lns[1].bias)
    want = math.sqrt(768) * float(lns[1].weight.abs().max()) + float(lns[1].bias.abs().max())
    assert abs(float(r1.max().view(torch.float32)) - want) <= 1e-5 * want
    # through the op: K12's output carries the bound, the Linear behind it uses it
    with switches.override(amax_hints=True, ln_bound_hints=True):
        x = _rand((2048, 192), 70)
        y = ops.add_layernorm(x, None, lns[0].weight, lns[0].bias)
        assert ops.amax_hint_get(y) is not None
        w = _rand((576, 192), 71, 0.05)
        out = ops.linear(y, w, None)
        ref = y.double() @ w.double().t()
        assert _err(out, ref) <= 2e-6


@gpu
def test_a_forward_without_a_hint_does_not_inherit_a_freed_tensors_record():
    """`linear()` re-attached `_LAST_HINT` by ADDRESS after `Function.apply`.  A forward that sets no
    hint (library path: fewer rows than gemm32s_min) whose output the caching allocator places at the address of an earlier,
    freed, hinted tensor must NOT come back carrying that tensor's absmax record — a 1e-4 record on an O(1) activation is an
    IEEE-half overflow inside the next K20 product."""
    from mask_bev_amd import ops
    lin = torch.nn.Linear(192, 1536).to(_dev())
    xs = _rand((512, 192), 5, 1.0)                           # 512 rows < gemm32s_min: the library's f32 GEMM, no hint
    hit = 0
    with switches.override(amax_hints=True):
        for attempt in range(8):
            x, w = _rand((4096, 192), 10 + attempt, 1e-4), _rand((192, 192), 20 + attempt, 1.0)
            h = ops.gemm32s_nt(x, w, None, hint_out=True)    # (4096, 192) f32 = the bytes of a (512, 1536) f32 output
            assert ops.amax_hint_get(h) is not None
            addr = h.data_ptr()
            del h
            y = ops.linear(xs, lin.weight, lin.bias)
            if y.data_ptr() == addr:
                hit += 1
                assert ops.amax_hint_get(y) is None, 'the output inherited the record of a freed tensor at its address'
            del y
    assert hit > 0, 'the allocator never reused the address: the test did not exercise the case'


@gpu
@pytest.mark.parametrize('graphed', [False, True])
def test_every_consumed_absmax_record_bounds_its_operand_over_a_whole_fp32_step(graphed, capsys):
    """VERDICT r05 #6a/b: `switches.amax_verify` — every record a K20 product (or K4's split mode) consumes during a whole
    fp32 training step of the BENCH workload is compared with a fresh max|operand| taken right in front of the product:
    producer hints, derived bounds (|gelu(z)| <= |z|, convex attention outputs, sqrt(C) max|gamma| + max|beta|), weight
    records keyed by the optimizer epoch, the registered record of the graph's static input.  Eager step, and the
    two-graph step (checks captured with the launches: they run again on every replay, here on a DIFFERENT batch than
    the capture saw, after an optimizer step rewrote the weights through raw pointers).
    record >= max|x| ALWAYS (a smaller one is an f16 overflow: the accuracy-critical direction).  Looseness only eats the
    18 binades of headroom under the record inside which an element keeps all 22 product bits: asserted <= 2^8
    (measured worst: 108x — a row slice of a shared K/V gradient bounded by the whole tensor's record; LayerNorm bounds
    sqrt(C) max|gamma| + max|beta| sit at 3-12x), i.e. >= 10 binades of full-precision range stay.  The loosest by call
    site are printed."""
    from mask_bev_amd import ops, synthetic
    from mask_bev_amd.graph import GraphedTrainStep
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    workload, batch = 'semantic_kitti_512', 2
    kw = synthetic.module_kwargs(workload, batch, compute_dtype='fp32')
    m = MaskBevModule(**kw).to(_dev()).train()
    m.log_scalars = False
    m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    data = [synthetic.make_batch(workload, batch, 0, s, _dev()) for s in range(3)]
    ops.AMAX_VERIFY.reset()
    with switches.override(amax_verify=True):
        if graphed:
            g = GraphedTrainStep(m, opt, data[0], warmup_iters=1)    # (its eager warm-up pass allocates the result buffer)
            for s in (1, 2, 0):
                g.step(data[s])                             # replays re-run the captured checks on new data / new weights
            rep = [e for e in ops.AMAX_VERIFY.report() if e[2]]      # the captured checks, as the LAST replay left them
            g.close()
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for s in range(2):
                    loss = m.training_step(data[s], s)
                    loss.backward()
                    opt.step()
                    del loss
            torch.cuda.current_stream().wait_stream(side)
            rep = ops.AMAX_VERIFY.report()
    ops.AMAX_VERIFY.reset()
    assert len(rep) > 100, len(rep)
    low = [(w, s, t, r) for w, s, c, t, r in rep if not (r >= t)]
    loose = sorted(((r / t, w, s) for w, s, c, t, r in rep if t > 0 and r > t), reverse=True)
    with capsys.disabled():
        print(f'\namax_verify ({"graph" if graphed else "eager"}): {len(rep)} consumed records checked, {len(low)} below the truth; '
              f'loosest: ' + '; '.join(f'{q:.1f}x {w} {s}' for q, w, s in loose[:6]))
    assert not low, low[:5]
    assert all(q <= 256.0 for q, _, _ in loose), loose[:5]
