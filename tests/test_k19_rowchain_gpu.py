"""K19 (csrc/rowchain.hip, mask_bev_amd/decoder_fused.py): row-local stage chains of the transformer decoder.

* every stage operation against an f64 torch expression of the same arithmetic (f32 weights: exact-f32 MFMA, 2e-5 of the
  operand scale; 16-bit weights: the activations are rounded to that type per GEMM — compared with the f64 product of the
  ROUNDED operands, so the tolerance is accumulation only);
* the fused decoder layers (_DecA / _DecB) inside the whole model against the unfused per-op path
  (MBV_DECODER_FUSED=0): outputs, loss and every parameter gradient; the oracle comparisons of test_model_gpu.py run
  through the fused path by default."""
import pytest
import torch
from mask_bev_amd import switches

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16, torch.float16]


def _r(shape, seed, scale=1.0, device='cuda'):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(device)


def _round(x, dt):
    return x if dt == torch.float32 else x.to(dt).float()


@pytest.mark.parametrize('wdt', DTS)
@pytest.mark.parametrize('rows,e,q', [(16, 128, 8), (400, 256, 100), (200, 256, 200), (37, 160, 37)])
def test_rowchain_forward_stages(wdt, rows, e, q):
    """LOAD (+ positions modulo q), GEMM (bias / ReLU / row and column offsets / accumulate over K chunks / narrow N),
    LN (+ residual, saved sum, stats), slot + positions, STORE in three dtypes."""
    from mask_bev_amd import decoder_fused as DF
    dev = torch.device('cuda', 0)
    f = 2 * e
    x = _r((rows, e), 1)
    o = _r((rows, e), 2).to(torch.bfloat16)
    pos = _r((q, e), 3)
    wo, bo = _r((e, e), 4, 0.1), _r((e,), 5)
    w1, b1 = _r((f, e), 6, 0.1), _r((f,), 7)
    w2, b2 = _r((e, f), 8, 0.1), _r((e,), 9)
    wc, bc = _r((2, e), 10, 0.1), _r((2,), 11)
    gam, bet = _r((e,), 12) + 1.5, _r((e,), 13)
    W = lambda t: t.to(wdt).contiguous()
    s_sum = torch.empty((rows, e), device=dev)
    stats = torch.empty((rows, 2), device=dev)
    y = torch.empty((rows, e), device=dev)
    hid = torch.empty((rows, f), device=dev)
    out = torch.empty((rows, e), device=dev)
    out16 = torch.empty((rows, e), dtype=torch.float16, device=dev)
    cls = torch.empty((rows, 2), device=dev)
    tq = torch.empty((rows, e), device=dev)
    P = DF.Program(rows, q, 1e-5, wdt)
    P.load(0, o, e)
    P.gemm(1, 0, W(wo), e, e, bias=bo)
    P.load(2, x, e)
    P.ln(3, 2, 1, gam, bet, e, stats=stats, save_sum=True)
    P.store(2, s_sum, e)
    P.store(3, y, e)
    w1c, w2c = W(w1), W(w2)
    if wdt != torch.float32 and e % 32 == 0:          # 16-bit programs: the fragment-major operand form (row / block offsets)
        w1c, w2c = DF.fragment_copy(w1c), DF.fragment_copy(w2c)
    ch = e                                # two K chunks of the hidden layer
    for c in range(0, f, ch):
        P.gemm(0, 3, w1c, ch, e, bias=b1, row0=c, bias0=c, relu=True, out=hid, out_col0=c)      # stored by the epilogue
        last = c + ch >= f
        P.gemm(1, 0, w2c, e, ch, bias=b2 if c == 0 else None, col0=c, accum=c > 0, out=out16 if last else None)
    P.store(1, out, e)
    P.gemm(4, 3, W(wc), 2, e, bias=bc)
    P.store(4, cls, 2)
    P.load_slot_plus(5, 3, pos, e)
    P.store(5, tq, e)
    P.run()
    torch.cuda.synchronize()
    # reference (f64 on the operands as the kernel sees them)
    D = torch.float64
    Wr = lambda t: _round(t, wdt).to(D)
    A = lambda t: _round(t, wdt).to(D)           # activations are rounded when they enter a 16-bit GEMM
    proj = A(o.float()) @ Wr(wo).t() + bo.to(D)
    ssum = x.to(D) + proj
    mean = ssum.mean(-1, keepdim=True)
    var = ((ssum - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    yr = (ssum - mean) * rstd * gam.to(D) + bet.to(D)
    tol = {torch.float32: 2e-5, torch.bfloat16: 2e-5, torch.float16: 2e-5}[wdt]
    sc = float(ssum.abs().max())
    assert float((s_sum.to(D) - ssum).abs().max()) < tol * sc * 4
    assert float((stats[:, 0].to(D) - mean[:, 0]).abs().max()) < 1e-5 * sc
    assert float((stats[:, 1].to(D) / rstd[:, 0] - 1).abs().max()) < 1e-4
    assert float((y.to(D) - yr).abs().max()) < 1e-4 * float(yr.abs().max())
    # downstream stages are checked on the kernel's own (f32) LayerNorm output
    yk = y.to(D)
    h = torch.relu(A(y) @ Wr(w1).t() + b1.to(D))
    assert float((hid.to(D) - h).abs().max()) < 1e-4 * float(h.abs().max())
    o2 = A(hid) @ Wr(w2).t() + b2.to(D)
    assert float((out.to(D) - o2).abs().max()) < 2e-4 * float(o2.abs().max())
    assert torch.equal(out16, out.to(torch.float16))
    c2 = A(y) @ Wr(wc).t() + bc.to(D)
    assert float((cls.to(D) - c2).abs().max()) < 1e-4 * float(c2.abs().max())
    idx = torch.arange(rows, device=dev) % q
    assert torch.equal(tq, y + pos[idx])
    del yk


@pytest.mark.parametrize('wdt', DTS)
@pytest.mark.parametrize('rows,e', [(16, 128), (400, 256), (90, 256)])
def test_rowchain_backward_stages(wdt, rows, e):
    """LN_BWD (+ per-block partial parameter gradients), GEMM against a transposed weight with the ReLU mask, COLSUM,
    ADD, accumulate-STORE."""
    from mask_bev_amd import decoder_fused as DF
    dev = torch.device('cuda', 0)
    f = e
    g = _r((rows, e), 21)
    ssum = _r((rows, e), 22, 2.0)
    gam = _r((e,), 23) + 1.5
    w2 = _r((e, f), 24, 0.1)                   # forward: y = h @ w2^T, h (rows, f)
    hmask = torch.relu(_r((rows, f), 25))
    acc0 = _r((rows, f), 26)
    D = torch.float64
    mean = ssum.to(D).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((ssum.to(D) - mean) ** 2).mean(-1, keepdim=True) + 1e-5)
    stats = torch.cat([mean, rstd], 1).float().contiguous()
    nblk = (rows + 15) // 16
    part_ln = torch.full((nblk, 2 * e), 9.0, device=dev)
    part_b = torch.full((nblk, e + f), 9.0, device=dev)
    ds = torch.empty((rows, e), device=dev)
    dh = torch.empty((rows, f), device=dev)
    acc = acc0.clone()
    w2t = w2.t().contiguous().to(wdt)            # (f, e): the transposed copy the data gradient streams
    P = DF.Program(rows, rows, 1e-5, wdt)
    P.load(0, g, e)
    P.load(1, ssum, e)
    P.ln_bwd(2, 0, 1, gam, stats, e, partial=part_ln)
    P.store(2, ds, e)
    P.colsum(2, part_b, e, 0)
    P.load(3, hmask, f)
    P.gemm(4, 2, w2t, f, e, mask=3)
    P.store(4, dh, f)
    P.colsum(4, part_b, f, e)
    P.add(5, 4, 3, f)
    P.store(5, acc, f, accum=True)
    P.run()
    torch.cuda.synchronize()
    xh = (ssum.to(D) - mean) * rstd
    gw = g.to(D) * gam.to(D)
    ds_r = rstd * (gw - gw.mean(-1, keepdim=True) - xh * (gw * xh).mean(-1, keepdim=True))
    assert float((ds.to(D) - ds_r).abs().max()) < 2e-5 * float(ds_r.abs().max()) * 4
    dgam, dbet = (g.to(D) * xh).sum(0), g.to(D).sum(0)
    got = part_ln.to(D).sum(0)
    assert float((got[:e] - dgam).abs().max()) < 1e-4 * float(dgam.abs().max())
    assert float((got[e:] - dbet).abs().max()) < 1e-4 * float(dbet.abs().max())
    dsk = _round(ds, wdt).to(D)
    dh_r = (dsk @ _round(w2, wdt).to(D)) * (hmask.to(D) > 0)
    assert float((dh.to(D) - dh_r).abs().max()) < 1e-4 * float(dh_r.abs().max())
    pb = part_b.to(D).sum(0)
    assert float((pb[:e] - ds.to(D).sum(0)).abs().max()) < 1e-4 * float(ds.abs().sum(0).max())
    assert float((pb[e:] - dh.to(D).sum(0)).abs().max()) < 1e-4 * float(dh.abs().sum(0).max())
    assert float((acc.to(D) - (acc0.to(D) + dh.to(D) + hmask.to(D))).abs().max()) < 1e-5 * 10


@pytest.mark.parametrize('wdt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('rows,e,f', [(400, 256, 2048), (16, 128, 1024), (90, 256, 512), (37, 160, 2048)])
def test_rowchain_ffn_stage(wdt, rows, e, f):
    """MBV_RC_FFN forward and backward (hidden chunks spread over the waves, partial outputs summed in an f64 LDS image)
    against f64 on the operands as the kernel rounds them: the input and the hidden image enter their GEMMs in the weight
    dtype; the hidden activations / their gradient leave as f32."""
    from mask_bev_amd import decoder_fused as DF
    dev = torch.device('cuda', 0)
    D = torch.float64
    x = _r((rows, e), 31)
    w1, b1 = _r((f, e), 32, 0.1), _r((f,), 33)
    w2, b2 = _r((e, f), 34, 0.05), _r((e,), 35)
    hid = torch.full((rows, f), -7.0, device=dev)
    y = torch.empty((rows, e), device=dev)
    P = DF.Program(rows, rows, 1e-5, wdt)
    P.load(0, x, e)
    P.ffn(1, 0, 2, DF.fragment_copy(w1.to(wdt)), DF.fragment_copy(w2.to(wdt)), e, f, hid, bias_a=b1, bias_out=b2)
    P.store(1, y, e)
    P.run()
    torch.cuda.synchronize()
    R = lambda t: _round(t, wdt).to(D)
    h_ref = torch.relu(R(x) @ R(w1).t() + b1.to(D))
    assert float((hid.to(D) - h_ref).abs().max()) < 1e-4 * float(h_ref.abs().max())
    y_ref = R(hid) @ R(w2).t() + b2.to(D)
    assert float((y.to(D) - y_ref).abs().max()) < 2e-4 * float(y_ref.abs().max())
    # backward: dh = (g W2) * (hid > 0), dx = dh W1, column partials of dh
    g = _r((rows, e), 36)
    nblk = (rows + 15) // 16
    dh = torch.full((rows, f), -7.0, device=dev)
    part = torch.full((nblk, f + 8), 9.0, device=dev)
    dx = torch.empty((rows, e), device=dev)
    # the data-gradient operands: fragment-major copies of W2^T (f, e) and W1^T (e, f) made from the row-major weights
    w2t, w1t = DF.fragment_copy(w2.to(wdt), transposed=True), DF.fragment_copy(w1.to(wdt), transposed=True)
    P = DF.Program(rows, rows, 1e-5, wdt)
    P.load(0, g, e)
    P.ffn(1, 0, 2, w2t, w1t, e, f, hid, backward=True, d_hid=dh, partial=part, partial_col0=8)
    P.store(1, dx, e)
    P.run()
    torch.cuda.synchronize()
    dh_ref = (R(g) @ R(w2)) * (hid.to(D) > 0)
    assert float((dh.to(D) - dh_ref).abs().max()) < 1e-4 * float(dh_ref.abs().max())
    dx_ref = R(dh) @ R(w1)
    assert float((dx.to(D) - dx_ref).abs().max()) < 2e-4 * float(dx_ref.abs().max())
    assert float((part[:, 8:].to(D).sum(0) - dh.to(D).sum(0)).abs().max()) < 1e-4 * float(dh.abs().sum(0).max())
    assert bool((part[:, :8] == 9.0).all())


@pytest.mark.parametrize('wdt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('rows,e,f', [(400, 256, 2048), (37, 160, 512), (16, 32, 256)])
def test_rowchain_split_launch(wdt, rows, e, f):
    """A split launch (f / 256 workgroups per row block: MBV_RC_SLICE, MBV_RC_SPLIT, stage owners) followed by the launch
    that adds the parts (MBV_RC_SUM) equals the one-workgroup MLP stage: same hidden activations bit for bit, the output up
    to the summation order of the partial products; forward and backward.  Stages with an owner run once: the ACCUM
    store of workgroup 1 adds exactly once, the GEMMs of workgroups 0 and 2 see their own operands (the prefetch chain
    steps over stages that belong to other workgroups)."""
    from mask_bev_amd import decoder_fused as DF
    dev = torch.device('cuda', 0)
    S = f // 256
    x = _r((rows, e), 41)
    w1, b1 = _r((f, e), 42, 0.1), _r((f,), 43)
    w2, b2 = _r((e, f), 44, 0.05), _r((e,), 45)
    wa, wb = _r((e, e), 46, 0.1).to(wdt), _r((e, e), 47, 0.1).to(wdt)
    fa, fb = DF.fragment_copy(wa), DF.fragment_copy(wb)
    w1c, w2c = DF.fragment_copy(w1.to(wdt)), DF.fragment_copy(w2.to(wdt))

    def one(split):
        hid = torch.full((rows, f), -7.0, device=dev)
        y = torch.empty((rows, e), device=dev)
        ga, gb = torch.empty((rows, e), device=dev), torch.empty((rows, e), device=dev)
        acc = torch.ones((rows, e), device=dev)
        if split == 1:
            P = DF.Program(rows, rows, 1e-5, wdt)
            P.load(0, x, e)
            P.gemm(2, 0, fa, e, e, out=ga)
            P.store(0, acc, e, accum=True)
            P.gemm(3, 0, fb, e, e, out=gb)
            P.ffn(1, 0, 2, w1c, w2c, e, f, hid, bias_a=b1, bias_out=b2)
            P.store(1, y, e)
            P.run()
        else:
            parts = torch.full((split, rows, e), float('nan'), device=dev)
            P = DF.Program(rows, rows, 1e-5, wdt, split=split)
            P.load(0, x, e)
            with P.only(0):
                P.gemm(2, 0, fa, e, e, out=ga)
            with P.only(min(1, split - 1)):
                P.store(0, acc, e, accum=True)
            with P.only(split - 1):
                P.gemm(3, 0, fb, e, e, out=gb)
            P.ffn(1, 0, 2, w1c, DF.fragment_copy(w2.to(wdt), kmajor=True), e, f, hid, bias_a=b1, bias_out=b2, sliced=True)
            P.store_part(1, parts, e)
            P.run()
            P = DF.Program(rows, rows, 1e-5, wdt)
            P.sum_parts(1, parts, e)
            P.store(1, y, e)
            P.run()
        torch.cuda.synchronize()
        return hid, y, ga, gb, acc

    ref, got = one(1), one(S)
    assert torch.equal(ref[0], got[0])                                   # hidden activations
    assert float((ref[1] - got[1]).abs().max()) < 1e-5 * float(ref[1].abs().max())
    assert torch.equal(ref[2], got[2]) and torch.equal(ref[3], got[3]) and torch.equal(ref[4], got[4])
    # backward
    hid = ref[0]
    g = _r((rows, e), 48)
    nblk = (rows + 15) // 16
    w2t, w1t = DF.fragment_copy(w2.to(wdt), transposed=True), DF.fragment_copy(w1.to(wdt), transposed=True)

    def bwd(split):
        dh = torch.full((rows, f), -7.0, device=dev)
        part = torch.full((nblk, f), 9.0, device=dev)
        csum = torch.full((nblk, e), 9.0, device=dev)
        dx = torch.empty((rows, e), device=dev)
        if split == 1:
            P = DF.Program(rows, rows, 1e-5, wdt)
            P.load(0, g, e)
            P.colsum(0, csum, e)
            P.ffn(1, 0, 2, w2t, w1t, e, f, hid, backward=True, d_hid=dh, partial=part)
            P.store(1, dx, e)
            P.run()
        else:
            parts = torch.full((split, rows, e), float('nan'), device=dev)
            P = DF.Program(rows, rows, 1e-5, wdt, split=split)
            P.load(0, g, e)
            with P.only(0):
                P.colsum(0, csum, e)
            P.ffn(1, 0, 2, w2t, DF.fragment_copy(w1.to(wdt), transposed=True, kmajor=True), e, f, hid, backward=True, d_hid=dh,
                  partial=part, sliced=True)
            P.store_part(1, parts, e)
            P.run()
            P = DF.Program(rows, rows, 1e-5, wdt)
            P.sum_parts(1, parts, e)
            P.store(1, dx, e)
            P.run()
        torch.cuda.synchronize()
        return dh, part, csum, dx

    ref, got = bwd(1), bwd(S)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2])
    assert float((ref[3] - got[3]).abs().max()) < 1e-5 * float(ref[3].abs().max())


def test_rowchain_rejects_bad_programs():
    from mask_bev_amd import decoder_fused as DF
    from mask_bev_amd._lib import MaskBevHipError
    x = torch.zeros((16, 64), device='cuda')
    w = torch.zeros((64, 64), device='cuda')
    P = DF.Program(16, 16, 1e-5, torch.float32)
    P.load(0, x, 64)
    P.gemm(0, 0, w, 64, 64)                  # in place
    with pytest.raises(MaskBevHipError):
        P.run()
    P = DF.Program(16, 16, 1e-5, torch.float32)
    P.load(9, x, 64)                         # no such slot
    with pytest.raises(MaskBevHipError):
        P.run()
    with pytest.raises(MaskBevHipError):
        DF.Program(16, 16, 1e-5, torch.bfloat16).gemm(1, 0, w, 64, 64)     # weight dtype != program dtype
    # an accumulating store of a split launch without an owner would be added once per workgroup of the row block:
    # refused by the host-side builder and, for callers of the C ABI, by the library (MBV_ERR_BAD_ARG)
    acc = torch.zeros((16, 64), device='cuda')
    P = DF.Program(16, 16, 1e-5, torch.float32, split=2)
    P.load(0, x, 64)
    with pytest.raises(MaskBevHipError):
        P.store(0, acc, 64, accum=True)
    with P.only(1):
        P.store(0, acc, 64, accum=True)          # owned: fine
    P.run()
    P = DF.Program(16, 16, 1e-5, torch.float32, split=2)
    P.load(0, x, 64)
    with P.only(0):
        P.store(0, acc, 64, accum=True)
    P.stages[-1].flags &= 0xff                   # strip the owner behind the builder's back
    with pytest.raises(MaskBevHipError):
        P.run()


def _fragment_layout(w):
    """dst[((t * KBN + kb) * 64 + lane) * 8 + j] = W[t * 16 + lane % 16][kb * 32 + 8 * (lane // 16) + j] (zero rows beyond W)."""
    rows, cols = w.shape
    tn, kbn = (rows + 15) // 16, cols // 32
    pad = torch.zeros((tn * 16, cols), dtype=w.dtype, device=w.device)
    pad[:rows] = w
    # (t, m, kb, g, j) -> (t, kb, g, m, j): lane = g * 16 + m
    return pad.view(tn, 16, kbn, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


@pytest.mark.parametrize('shape', [(256, 256), (2048, 256), (256, 2048), (10, 64), (70, 96), (3, 32), (129, 160)])
def test_fragment_group_layout(shape):
    """Both orientations of the fragment-major copies, incl. matrices that are not whole 64 x 64 tiles, unaligned row
    pitches and sub-matrix views (the scalar-load path of the kernel)."""
    from mask_bev_amd import decoder_fused as DF
    r, c = shape
    w = _r((r, c), 5).to(torch.bfloat16)
    assert torch.equal(DF.fragment_copy(w).t, _fragment_layout(w))
    tn, kbn = (r + 15) // 16, c // 32                            # k-major: block (t, kb) at kb * tn + t
    km = _fragment_layout(w).view(tn, kbn, 512).transpose(0, 1).reshape(-1)
    assert torch.equal(DF.fragment_copy(w, kmajor=True).t, km)
    wt = _r((c, r), 6).to(torch.bfloat16)                      # logical W = wt^T
    assert torch.equal(DF.fragment_copy(wt, transposed=True).t, _fragment_layout(wt.t().contiguous()))
    big = _r((r + 3, c + 5), 7).to(torch.bfloat16)             # a view with an odd pitch and an odd start
    v = big[1:1 + r, 3:3 + c]
    assert torch.equal(DF.fragment_copy(v).t, _fragment_layout(v.contiguous()))
    bigt = _r((c + 1, r + 7), 8).to(torch.bfloat16)
    vt = bigt[1:1 + c, 5:5 + r]
    assert torch.equal(DF.fragment_copy(vt, transposed=True).t, _fragment_layout(vt.t().contiguous()))


def test_transpose_group():
    import ctypes
    from mask_bev_amd import _lib, ops
    lib = _lib.load()
    for dt, es in ((torch.float32, 4), (torch.bfloat16, 2)):
        srcs = [_r((70, 130), 1).to(dt), _r((256, 2048), 2).to(dt), _r((1, 5), 3).to(dt)]
        dsts = [torch.empty((s.shape[1], s.shape[0]), dtype=dt, device='cuda') for s in srcs]
        n = len(srcs)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
        ops.check(lib.mbv_transpose_group(PA(*[s.data_ptr() for s in srcs]), PA(*[d.data_ptr() for d in dsts]),
                                          IA(*[s.shape[0] for s in srcs]), IA(*[s.shape[1] for s in srcs]), n, es,
                                          ops._stream()), 'mbv_transpose_group')
        for s, d in zip(srcs, dsts):
            assert torch.equal(d, s.t())


def _run_model(device, dtype, fused, monkeypatch, arena):
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from oracle import maskbev_oracle as O
    from tests.util_cfg import random_gt, random_scans, tiny_kwargs
    switches.patch(monkeypatch, decoder_fused='1' if fused else '0')
    okw = tiny_kwargs()
    kw = dict(okw, compute_dtype=dtype)
    sd = O.make_state_dict(O.make_cfg(**okw), 7)
    m = MaskBevModule(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).train()
    if arena:
        m.flatten_parameters()
    head = m._panoptic_head._panoptic_head
    head.num_points = 256
    head.point_seed = 11
    scans = [s.to(device) for s in random_scans(okw, [3000, 2000], seed=2)]
    labels, gt = random_gt(okw, 2, 3, seed=4)
    with torch.no_grad():
        cls, masks, _ = m(scans)
    loss = m.training_step((scans, (labels.to(device), gt.to(device))), 1)
    m.scale_loss(loss).backward()
    from mask_bev_amd import ops
    ops.flush_deferred_grads()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    return [c.float() for c in cls], [mk.float() for mk in masks], float(loss.detach()), grads


@pytest.mark.parametrize('dtype,arena', [('fp32', False), ('fp32', True), ('bf16', True)])
def test_fused_decoder_equals_unfused_path(device, monkeypatch, dtype, arena):
    """Whole model, same weights / inputs / sampling points, decoder query side fused (K19) vs per-op: fp32 — outputs
    1e-4, loss 1e-5, every parameter gradient 2e-3 of its largest entry (exact-f32 MFMA both ways, other summation
    orders); bf16 — the fused chain rounds the decoder's GEMM operands to bf16 where the per-op path kept its few-row
    Linears in f32, so only the declared 16-bit tolerances of test_model_gpu.py apply (6e-2 / 3e-1)."""
    cls_f, mk_f, loss_f, g_f = _run_model(device, dtype, True, monkeypatch, arena)
    cls_u, mk_u, loss_u, g_u = _run_model(device, dtype, False, monkeypatch, arena)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-9))
    to, tl, tg = (1e-4, 1e-5, 2e-3) if dtype == 'fp32' else (6e-2, 2e-2, 3e-1)
    for i in range(10):
        assert rel(mk_f[i], mk_u[i]) < (to if dtype == 'fp32' or i == 9 else 3e-1), i
        assert rel(cls_f[i], cls_u[i]) < (to if dtype == 'fp32' else 3e-1), i
    assert abs(loss_f - loss_u) / abs(loss_u) < tl
    assert g_f.keys() == g_u.keys()
    worst = max((rel(g_f[k], g_u[k]), k) for k in g_u)
    assert worst[0] < tg, worst


def test_no_grad_forward_between_training_forward_and_backward(device):
    """A forward under no_grad through the same head between a training forward and its backward (a sanity validation,
    the bench's extra eager steps) refreshes the chains' weight copies WITHOUT the transposed operands; the pending
    backward must still find its own (WeightCopies.refresh used to wipe them -> KeyError)."""
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from tests.util_cfg import random_gt, random_scans, tiny_kwargs
    okw = tiny_kwargs()
    torch.manual_seed(3)
    m = MaskBevModule(**dict(okw, compute_dtype='bf16')).to(device).train()
    m.flatten_parameters()
    scans = [s.to(device) for s in random_scans(okw, [3000, 2000], seed=2)]
    labels, gt = random_gt(okw, 2, 3, seed=4)
    batch = (scans, (labels.to(device), gt.to(device)))
    loss = m.training_step(batch, 0)
    with torch.no_grad():
        m(scans)                                   # the interleaved inference pass
    m.scale_loss(loss).backward()
    from mask_bev_amd import ops
    ops.flush_deferred_grads()
    torch.cuda.synchronize()
    g = m._panoptic_head._panoptic_head.transformer_decoder.layers[0].ffn.layers[1].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_fragment_group_more_entries_than_one_launch_takes():
    """mbv_fragment_group splits its entry list into launches of MBV_TR_MAX (96): 130 small matrices of mixed shapes and
    orientations in one call (the decoder refreshes 130 copies per step)."""
    import ctypes
    from mask_bev_amd import _lib, ops
    lib = _lib.load()
    srcs, wants, rows_l, cols_l, tr_l = [], [], [], [], []
    for i in range(130):
        r, c = 16 * (1 + i % 5) + (i % 3), 32 * (1 + i % 4)
        tr = i % 2
        w = _r((c, r) if tr else (r, c), 100 + i).to(torch.bfloat16)
        srcs.append(w)
        wants.append(_fragment_layout(w.t().contiguous() if tr else w))
        rows_l.append(r); cols_l.append(c); tr_l.append(tr)
    dsts = [torch.empty_like(w) for w in wants]
    n = len(srcs)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
    ops.check(lib.mbv_fragment_group(PA(*[s.data_ptr() for s in srcs]), PA(*[d.data_ptr() for d in dsts]), IA(*rows_l),
                                     IA(*cols_l), IA(*[s.stride(0) for s in srcs]), IA(*tr_l), n, ops._stream()),
              'mbv_fragment_group')
    torch.cuda.synchronize()
    for i, (d, w) in enumerate(zip(dsts, wants)):
        assert torch.equal(d, w), i


def test_split_launch_with_an_owned_load_stage():
    """A LOAD stage (slot + positional rows) owned by one workgroup of a split launch: the others step over it and the
    operand prefetch chain (next LOAD / next GEMM requested one stage ahead) still hands every stage its own operands."""
    from mask_bev_amd import decoder_fused as DF
    dev = torch.device('cuda', 0)
    rows, e, q, wdt = 90, 64, 30, torch.bfloat16
    x, pos, x2 = _r((rows, e), 61), _r((q, e), 62), _r((rows, e), 63)
    fa, fb = DF.fragment_copy(_r((e, e), 64, 0.1).to(wdt)), DF.fragment_copy(_r((e, e), 65, 0.1).to(wdt))

    def run(split):
        a, b, c = (torch.empty((rows, e), device=dev) for _ in range(3))
        P = DF.Program(rows, q, 1e-5, wdt, split=split)
        P.load(0, x, e)
        with P.only(0):
            P.load_slot_plus(1, 0, pos, e)          # owned LOAD (slot 0 + positional rows at r % q)
            P.gemm(2, 1, fa, e, e, out=a)
        P.load(3, x2, e)                            # a LOAD every workgroup runs, behind the owned one
        with P.only(1):
            P.gemm(4, 3, fb, e, e, out=b)
        with P.only(2):
            P.gemm(5, 0, fb, e, e, out=c)
        P.run()
        torch.cuda.synchronize()
        return a, b, c

    ref, got = run(1), run(3)
    assert all(torch.equal(u, v) for u, v in zip(ref, got))
