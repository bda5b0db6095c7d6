"""K6 decoder attention (split-L forward + combine, flash-style backward) vs dense f32 attention.
f32 path: rtol 1e-4; bf16 path on bf16-rounded inputs: 2e-2 / 3e-2 (declared bf16 tolerance)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, blocked, heads):
    b, nq, e = q.shape
    nl = k.shape[1]
    d = e // heads
    qh = q.view(b, nq, heads, d).transpose(1, 2)
    kh = k.view(b, nl, heads, d).transpose(1, 2)
    vh = v.view(b, nl, heads, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / d ** 0.5
    if blocked is not None:
        s = s.masked_fill(blocked.view(b, 1, nq, nl), float('-inf'))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(b, nq, e)


CASES = [(2, 100, 1024, 8, 32, True), (1, 100, 100, 8, 32, False), (2, 8, 25, 8, 16, True), (1, 130, 300, 4, 16, True),
         (2, 100, 4096, 8, 32, True), (1, 6, 7, 2, 64, False)]


@pytest.mark.parametrize('B,Q,L,heads,D,masked', CASES)
@pytest.mark.parametrize('dtype', ['f32', 'f32_exact', 'bf16', 'fp16'])
def test_attention_fwd_bwd(device, B, Q, L, heads, D, masked, dtype):
    """'f32': the split mode (IEEE-half pairs on the 16-bit matrix pipe, per-tile scales); 'f32_exact': v_mfma_f32_32x32x2_f32
    (switches.k6_split off)."""
    from mask_bev_amd import ops, switches
    if dtype == 'f32_exact':
        with switches.override(k6_split=False):
            return test_attention_fwd_bwd(device, B, Q, L, heads, D, masked, 'f32')
    g = torch.Generator().manual_seed(Q * 7 + L)
    E = heads * D
    q, k, v = (torch.randn(B, n, E, generator=g) for n in (Q, L, L))
    go = torch.randn(B, Q, E, generator=g)
    blocked = None
    if masked:
        blocked = torch.rand(B, Q, L, generator=g) < 0.7
        blocked[:, 0] = True
        blocked[:, 0, L // 2] = False                      # a row with a single attendable key
        blocked[:, 1, :min(L, 128)] = True                 # a whole first split blocked (if L > 128 other keys remain)
        blocked[:, 1, -1] = False
    tdt = dict(f32=torch.float32, bf16=torch.bfloat16, fp16=torch.float16)[dtype]
    q, k, v, go = (t.to(tdt).float() for t in (q, k, v, go))
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    out_ref = ref_attention(qr, kr, vr, blocked, heads)
    out_ref.backward(go)
    qd, kd, vd = (t.to(device=device, dtype=tdt).requires_grad_() for t in (q, k, v))
    bd = None if blocked is None else blocked.to(device).unsqueeze(1)
    out = ops.attention(qd, kd, vd, bd, heads)
    out.backward(go.to(device=device, dtype=tdt))
    # declared tolerances: f32 exact-MFMA path; bf16 (8 significand bits); IEEE half (11 bits): a quarter of bf16's
    tol = {'f32': dict(rtol=1e-4, atol=2e-5), 'bf16': dict(rtol=2e-2, atol=2e-2), 'fp16': dict(rtol=5e-3, atol=5e-3)}[dtype]
    gtol = {'f32': dict(rtol=2e-4, atol=1e-4), 'bf16': dict(rtol=3e-2, atol=6e-2),
            'fp16': dict(rtol=8e-3, atol=1.5e-2)}[dtype]
    torch.testing.assert_close(out.detach().float().cpu(), out_ref.detach(), **tol)
    torch.testing.assert_close(qd.grad.float().cpu(), qr.grad, **gtol)
    torch.testing.assert_close(kd.grad.float().cpu(), kr.grad, **gtol)
    torch.testing.assert_close(vd.grad.float().cpu(), vr.grad, **gtol)


@pytest.mark.gpu
@pytest.mark.parametrize('arena', [False, True])
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_shared_kv_attention_equals_separate_projections(device, dt, arena):
    """ops.shared_kv_project + attention_shared_kv (one key GEMM and one value GEMM for three layers, K6 reading and
    writing columns of the shared matrices in place) vs three independent linear → attention chains: outputs and the
    gradients of the memory, the queries and the packed in_proj parameters."""
    from mask_bev_amd import ops
    torch.manual_seed(11)
    B, Q, L, E, H, n = 2, 100, 300, 64, 4, 3
    key_in = torch.randn(B, L, E, device=device).to(dt)
    val_in = torch.randn(B, L, E, device=device).to(dt)
    ws = [(0.2 * torch.randn(3 * E, E, device=device)).requires_grad_() for _ in range(n)]
    bs = [(0.1 * torch.randn(3 * E, device=device)).requires_grad_() for _ in range(n)]
    qs = [torch.randn(B, Q, E, device=device) for _ in range(n)]
    blocked = torch.rand(B, 1, Q, L, device=device) > 0.7
    blocked[:, :, :, 0] = False
    gos = [torch.randn(B, Q, E, device=device) for _ in range(n)]

    def run(shared):
        k_in, v_in = key_in.clone().requires_grad_(), val_in.clone().requires_grad_()
        q_l = [q.clone().requires_grad_() for q in qs]
        for p in ws + bs:
            p.grad = None
            p._mbv_arena = False
        if shared and arena:
            # arena parameters: f32 gradients that exist before the pass and ACCUMULATE (start them at 1.0) — the k / v rows
            # take their products in place, the 16-bit ones through the grouped launch at the end of the pass
            for p in ws + bs:
                p.grad = torch.ones_like(p, dtype=torch.float32)
                p._mbv_arena = True
        outs = []
        if shared:
            holder, token = ops.shared_kv_project(k_in, v_in, list(zip(ws, bs)))
            for j in range(n):
                outs.append(ops.attention_shared_kv(q_l[j], token, blocked, H, holder, j))
        else:
            for j in range(n):
                k = torch.nn.functional.linear(k_in, ws[j][E:2 * E].to(dt), bs[j][E:2 * E].to(dt))
                v = torch.nn.functional.linear(v_in, ws[j][2 * E:].to(dt), bs[j][2 * E:].to(dt))
                outs.append(ops.attention(q_l[j], k, v, blocked, H))
        torch.autograd.backward(outs, [g.to(o.dtype) for g, o in zip(gos, outs)])
        torch.cuda.synchronize()
        off = 1.0 if (shared and arena) else 0.0
        return ([o.detach().float() for o in outs], k_in.grad.float(), v_in.grad.float(), [q.grad.float() for q in q_l],
                [w.grad.clone() - off for w in ws], [b.grad.clone() - off for b in bs])

    a, b = run(True), run(False)
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    for x, y in zip(a[0], b[0]):
        torch.testing.assert_close(x, y, **tol)
    for idx in (1, 2):
        scale = float(b[idx].abs().max())
        torch.testing.assert_close(a[idx] / scale, b[idx] / scale, **tol)
    for idx in (3, 4, 5):
        for x, y in zip(a[idx], b[idx]):
            scale = float(y.abs().max()) + 1e-12
            torch.testing.assert_close(x / scale, y / scale, **tol)
    # the q rows of the packed parameters receive nothing from this path
    assert all(float(w[:E].abs().max()) == 0.0 for w in a[4])


@pytest.mark.gpu
def test_shared_kv_attention_bench_shape_bf16(device):
    """The bench's level-0 cross-attention (B 4, Q 100, L 16 384, E 256, 8 heads, bf16): slot reads / slot-wise bf16
    gradient stores of the shared key / value matrices vs the contiguous path, on the same projected values."""
    from mask_bev_amd import ops
    torch.manual_seed(5)
    B, Q, L, E, H, n = 4, 100, 16384, 256, 8, 3
    k_cat = torch.randn(B, L, n * E, device=device).bfloat16()
    v_cat = torch.randn(B, L, n * E, device=device).bfloat16()
    blocked = torch.rand(B, 1, Q, L, device=device) > 0.5
    blocked[..., :4] = False
    holder = ops.SharedKV()
    holder.k_cat, holder.v_cat, holder.n, holder.e = k_cat, v_cat, n, E
    token = torch.zeros((), device=device, requires_grad=True)
    for slot in (0, 2):
        q = torch.randn(B, Q, E, device=device, requires_grad=True)
        go = torch.randn(B, Q, E, device=device)
        out = ops.attention_shared_kv(q, token, blocked, H, holder, slot)
        out.backward(go.to(out.dtype))
        k = k_cat[..., slot * E:(slot + 1) * E].contiguous().requires_grad_()
        v = v_cat[..., slot * E:(slot + 1) * E].contiguous().requires_grad_()
        q2 = q.detach().clone().requires_grad_()
        ref = ops.attention(q2, k, v, blocked, H)
        ref.backward(go.to(ref.dtype))
        assert torch.equal(out, ref)                                   # same kernel, same values: bit-identical
        torch.testing.assert_close(q.grad, q2.grad, rtol=1e-4, atol=1e-6)   # split-L partial sums meet in f32 atomics
        dk = holder.dk_cat[..., slot * E:(slot + 1) * E].float()
        dv = holder.dv_cat[..., slot * E:(slot + 1) * E].float()
        assert torch.equal(dk, k.grad.float().bfloat16().float())      # the f32 gradient rounded to bf16 once
        assert torch.equal(dv, v.grad.float().bfloat16().float())
    # the untouched slot of the shared gradient matrices was never written by slots 0 and 2
    assert holder.written == {0, 2}


@pytest.mark.gpu
@pytest.mark.parametrize('arena', [False, True])
def test_level_inputs_node_on_the_gpu(device, arena):
    """ops.level_inputs in the compute dtype against the plain ops (flatten + level-embedding row + positions, two casts):
    values, the memory gradient, and the embedding-row gradient — returned (plain parameter) or accumulated into the arena
    gradient's row through the grouped column sums."""
    from mask_bev_amd import ops
    torch.manual_seed(4)
    mem = torch.randn(2, 32, 6, 10, device=device, requires_grad=True)
    lw = torch.randn(3, 32, device=device, requires_grad=True)
    pos = torch.randn(1, 60, 32, device=device)
    dt = torch.bfloat16
    ga, gk = torch.randn(2, 60, 32, device=device).to(dt), torch.randn(2, 60, 32, device=device).to(dt)
    if arena:
        lw.grad = torch.ones_like(lw)
        lw._mbv_arena = True
    a, k = ops.level_inputs(mem, lw, 2, pos, dt)
    torch.autograd.backward([a, k], [ga, gk])
    torch.cuda.synchronize()
    g_mem, g_lw = mem.grad.clone(), lw.grad.clone() - (1.0 if arena else 0.0)
    mem.grad = None
    lw2 = lw.detach().clone().requires_grad_()
    a2 = mem.flatten(2).transpose(1, 2) + lw2[2].view(1, 1, -1)
    k2 = a2 + pos
    torch.autograd.backward([a2.to(dt), k2.to(dt)], [ga, gk])
    assert torch.equal(a, a2.to(dt)) and torch.equal(k, k2.to(dt))
    assert torch.allclose(g_mem, mem.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(g_lw, lw2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Q,L,heads,D,gscale,kmax', [(2, 100, 1024, 8, 32, 1e-6, 1.0), (1, 100, 100, 8, 32, 1.0, 1.0),
                                                        (2, 100, 4096, 8, 32, 1e3, 1.0), (2, 100, 4096, 8, 32, 1e3, 10.0),
                                                        (1, 130, 300, 4, 16, 3e-4, 1.0), (1, 60, 200, 2, 64, 1.0, 1.0)])
def test_split_mode_against_float64(device, B, Q, L, heads, D, gscale, kmax):
    """K6's split mode (f32 tensors, products from IEEE-half pairs, per-tile power-of-two scales) against the float64 attention,
    with the exact-f32 MFMA form beside it: output and the three gradients within 4e-6 of the result's maximum — or, where f32
    arithmetic itself is the limit, within 2x the exact form's own error — for output gradients of 1e-6 ... 1e+3 and keys /
    values of very different magnitudes per head.  (A query with ONE attendable key has dS = p (dP - delta) = 0 by cancellation:
    both forms leave ~ 2-4e-5 of noise in dK there.  `kmax` = 10 makes logits of +- 40: a probability's exponent carries its
    dot product's error, 2^-21 of sum|q_d k_d| for half pairs against 2^-24 for f32 — 8x, inside the same bar.)"""
    from mask_bev_amd import ops, switches
    g = torch.Generator().manual_seed(Q + L + D)
    E = heads * D
    q = torch.randn(B, Q, E, generator=g)
    k = torch.randn(B, L, E, generator=g) * torch.logspace(-2, float(torch.log10(torch.tensor(kmax))), heads).repeat_interleave(D)
    v = torch.randn(B, L, E, generator=g) * torch.logspace(1, -3, heads).repeat_interleave(D)
    go = torch.randn(B, Q, E, generator=g) * gscale
    blocked = torch.rand(B, Q, L, generator=g) < 0.5
    blocked[:, 0] = True
    blocked[:, 0, L // 2] = False
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
    ref = ref_attention(qr, kr, vr, blocked, heads)
    ref.backward(go.double())

    def err(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp(min=1e-300))

    errs = {}
    for split in (False, True):
        with switches.override(k6_split=split):
            qd, kd, vd = (t.to(device).requires_grad_() for t in (q, k, v))
            out = ops.attention(qd, kd, vd, blocked.to(device).unsqueeze(1), heads)
            out.backward(go.to(device))
        errs[split] = {name: err(a, b) for name, a, b in (('out', out.detach(), ref.detach()), ('dq', qd.grad, qr.grad),
                                                           ('dk', kd.grad, kr.grad), ('dv', vd.grad, vr.grad))}
    for name, e in errs[True].items():
        bar = max(4e-6, 2.0 * errs[False][name]) * (8.0 if kmax > 1 else 1.0)
        assert e <= bar, (name, e, errs[False][name])
