"""K11 parameter arena: the flat AdamW / Adam kernel vs torch.optim (f32 rounding), the bias-gradient column sum,
the direct weight-gradient accumulation of ops.linear, and the whole module trained through the arena."""
import pytest
import torch

from tests.util_cfg import random_gt, random_scans, tiny_kwargs
from mask_bev_amd import switches

pytestmark = pytest.mark.gpu


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(37, 19)
        self.b = torch.nn.Linear(19, 5)
        self.n = torch.nn.LayerNorm(19)


@pytest.mark.parametrize('decoupled', [True, False])
def test_flat_adam_matches_torch(device, decoupled):
    from mask_bev_amd.arena import FlatAdam, ParameterArena
    torch.manual_seed(1)
    ref = _Toy().to(device)
    mine = _Toy().to(device)
    mine.load_state_dict(ref.state_dict())
    arena = ParameterArena([('encoder', mine.a), ('backbone', mine.n), ('head', mine.b)])
    assert arena.intact()
    groups = [dict(segment='encoder', lr=3e-3), dict(segment='backbone', lr=3e-3), dict(segment='head', lr=1e-2)]
    opt = FlatAdam(arena, groups, lr=3e-3, weight_decay=0.05, decoupled=decoupled)
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    topt = cls([dict(params=list(ref.a.parameters()) + list(ref.n.parameters()), lr=3e-3),
                dict(params=ref.b.parameters(), lr=1e-2)], lr=3e-3, weight_decay=0.05)
    g = torch.Generator(device='cpu').manual_seed(5)
    for it in range(6):
        for (n1, p), (n2, q) in zip(ref.named_parameters(), mine.named_parameters()):
            gr = torch.randn(p.shape, generator=g).to(device)
            p.grad = gr.clone()
            q.grad.copy_(gr)
        topt.step()
        opt.step()
        for (n1, p), (n2, q) in zip(ref.named_parameters(), mine.named_parameters()):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), (it, n1, (p - q).abs().max())
            assert float(q.grad.abs().max()) == 0.0                     # cleared by the same pass
            assert torch.equal(q._mbv_shadow, q.detach().to(torch.bfloat16))   # shadow = RNE bf16 of the new value


@pytest.mark.parametrize('T,N,dt', [(400, 256, torch.bfloat16), (65536, 384, torch.bfloat16), (1000, 2, torch.float32),
                                    (777, 131, torch.bfloat16), (5000, 1024, torch.float32), (65536, 384, torch.float16),
                                    (777, 131, torch.float16)])
def test_colsum_accum(device, T, N, dt):
    from mask_bev_amd import ops
    g = torch.randn(T, N, device=device).to(dt)
    out = torch.randn(N, device=device)
    want = out.double() + g.double().sum(0)
    ops.colsum_accum(g, out)
    assert torch.allclose(out.double(), want, rtol=1e-5, atol=1e-3 * (T ** 0.5) * 1e-2 + 1e-4)


@pytest.mark.parametrize('dt', [None, torch.bfloat16])
def test_linear_direct_grad_matches_autograd(device, dt):
    """Same Linear stack with and without the arena: outputs and parameter gradients agree."""
    from mask_bev_amd import ops
    from mask_bev_amd.arena import ParameterArena
    torch.manual_seed(2)
    ref, mine = _Toy().to(device), _Toy().to(device)
    mine.load_state_dict(ref.state_dict())
    arena = ParameterArena([('all', mine)], shadow_dtype=torch.bfloat16)

    def run(m, x):
        with torch.autocast('cuda', dtype=dt or torch.bfloat16, enabled=dt is not None):
            h = ops.linear(x, m.a.weight, m.a.bias)
            h = m.n(h.float())
            y = ops.linear(h, m.b.weight, m.b.bias, rows=(1, 4)) + ops.linear(h, m.b.weight, m.b.bias, rows=(0, 3))
        return y

    x = torch.randn(9000, 37, device=device)          # > 8192 rows: exercises the split-K weight gradient
    y0 = run(ref, x)
    y0.float().square().mean().backward()
    y1 = run(mine, x)
    y1.float().square().mean().backward()
    tol = 1e-5 if dt is None else 3e-2
    assert torch.allclose(y0.float(), y1.float(), rtol=tol, atol=tol)
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert q.grad.data_ptr() >= arena.grad.data_ptr()             # still the arena view
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((p.grad - q.grad).abs().max()) <= tol * scale, n


def test_module_trains_through_arena(device):
    """flatten_parameters(): same state_dict keys, same loss as the per-tensor module, parameters move, shadows and
    arena stay intact over optimizer steps; load_state_dict refreshes the shadow."""
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    kw['compute_dtype'] = 'bf16'
    kw['optimiser_type'] = 'adam_w'
    kw['lr'] = 5e-4
    m = MaskBevModule(**kw).to(device).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 2000
    keys_before = list(m.state_dict().keys())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    arena = m.flatten_parameters()
    assert list(m.state_dict().keys()) == keys_before
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    opt = m.configure_optimizers()['optimizer']
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    batch = (scans, (labels.to(device), gt.to(device)))
    losses = []
    for it in range(4):
        loss = m.training_step(batch, it)
        loss.backward()
        assert float(arena.grad.abs().sum()) > 0
        opt.step()
        opt.zero_grad()
        assert float(arena.grad.abs().max()) == 0.0
        losses.append(float(loss))
    assert arena.intact()
    assert all(l == l and abs(l) < 1e6 for l in losses)
    assert min(losses[1:]) < losses[0]
    w = m._backbone._backbone.stages[0].blocks[0].ffn.layers[0][0].weight
    assert not torch.equal(w, sd['_backbone._backbone.stages.0.blocks.0.ffn.layers.0.0.weight'])
    assert torch.equal(w._mbv_shadow, w.detach().to(torch.bfloat16))
    m.load_state_dict(sd)
    assert torch.equal(w._mbv_shadow, sd['_backbone._backbone.stages.0.blocks.0.ffn.layers.0.0.weight'].to(torch.bfloat16))


def test_graph_step_with_arena(device):
    """HIP-graph step over the arena: gradients accumulate into the static arena buffer inside the replay, the K11
    step clears them, and the loss goes down over a few steps on a fixed batch."""
    from mask_bev_amd.graph import GraphedTrainStep
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    kw['compute_dtype'] = 'bf16'
    kw['lr'] = 5e-4
    m = MaskBevModule(**kw).to(device).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 2000
    arena = m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    batch = (scans, (labels.to(device), gt.to(device)))
    g = GraphedTrainStep(m, opt, batch)
    p0 = arena.param.clone()
    losses = []
    for it in range(6):
        losses.append(float(g.step(batch)))
        assert float(arena.grad.abs().max()) == 0.0
    torch.cuda.synchronize()
    assert arena.intact()
    assert all(l == l for l in losses), losses
    assert min(losses[2:]) < losses[0], losses
    a, b = arena.segments['encoder']
    assert float((arena.param[a:b] - p0[a:b]).abs().max()) > 0          # encoder (eager) parameters move
    a, b = arena.segments['head']
    assert float((arena.param[a:b] - p0[a:b]).abs().max()) > 0          # graph-owned parameters move
    g.close()


@pytest.mark.parametrize('T,O,I', [(400, 256, 256), (400, 2048, 256), (400, 2, 256), (801, 256, 2048), (37, 19, 5),
                                   (2048, 96, 130)])
def test_wgrad_small_f32(device, T, O, I):
    from mask_bev_amd import _lib
    lib = _lib.load()
    g = torch.randn(T, O, device=device)
    x = torch.randn(T, I, device=device)
    acc = torch.randn(O, I, device=device)
    bacc = torch.randn(O, device=device)
    want = acc.double() + g.double().t() @ x.double()
    bwant = bacc.double() + g.double().sum(0)
    rc = lib.mbv_wgrad_small_f32(g.data_ptr(), x.data_ptr(), T, O, I, acc.data_ptr(), bacc.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.allclose(acc.double(), want, rtol=1e-5, atol=2e-5 * T ** 0.5)
    assert torch.allclose(bacc.double(), bwant, rtol=1e-5, atol=2e-5 * T ** 0.5)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
def test_arena_gradients_equal_plain_autograd(device, dtype):
    """Every direct-accumulation path at once (Linear dW / db incl. the small-token kernel, the bias gradients
    deferred to K12, K12's and K3's affine gradients): a flattened module and a plain one with the same weights,
    batch and sampling points produce the same loss and the same gradient for every parameter."""
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    kw['compute_dtype'] = dtype
    torch.manual_seed(0)
    plain = MaskBevModule(**kw).to(device).train()
    torch.manual_seed(0)
    flat = MaskBevModule(**kw).to(device).train()
    flat.load_state_dict(plain.state_dict())
    arena = flat.flatten_parameters()
    for m in (plain, flat):
        m.log_scalars = False
        h = m._panoptic_head._panoptic_head
        h.num_points = 600
        h.point_seed = 5                              # same sampling points in both evaluations
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    batch = (scans, (labels.to(device), gt.to(device)))
    ls = 1024.0 if dtype == 'fp16' else 1.0           # fp16: a fixed loss scale on both sides keeps gradients normal
    l0 = plain.training_step(batch, 0)
    (l0 * ls).backward()
    l1 = flat.training_step(batch, 0)
    (l1 * ls).backward()
    tol = 2e-4 if dtype == 'fp32' else 4e-2
    assert abs(float(l0) - float(l1)) <= tol * abs(float(l0))
    worst = []
    for (n, p), (_, q) in zip(plain.named_parameters(), flat.named_parameters()):
        assert (p.grad is None) == (q.grad is None or float(q.grad.abs().max()) == 0.0 and p.grad is None), n
        if p.grad is None:
            continue
        scale = float(p.grad.abs().max()) + 1e-9
        err = float((p.grad.float() - q.grad).abs().max()) / scale
        worst.append((err, n))
        assert err <= (5e-3 if dtype == 'fp32' else 0.15), (n, err)
    assert max(worst)[0] < (5e-3 if dtype == 'fp32' else 0.15)


@pytest.mark.parametrize('kind', ['gelu', 'relu'])
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_bias_act_backward(device, kind, dt):
    """Fused activation backward + bias column sums vs torch."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(3000, 96, generator=g) * 2).to(device).to(dt).requires_grad_()
    bias = torch.nn.Parameter(torch.zeros(96, device=device))
    bias.grad = torch.randn(96, generator=g).to(device)
    b0 = bias.grad.clone()
    go = torch.randn(3000, 96, generator=g).to(device).to(dt)
    y = ops.bias_act(z, bias, kind)
    y.backward(go)
    zr = z.detach().double().requires_grad_()
    yr = torch.nn.functional.gelu(zr) if kind == 'gelu' else torch.relu(zr)
    yr.backward(go.double())
    tol = {torch.float32: 1e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dt]
    assert torch.allclose(y.double(), yr, rtol=tol, atol=tol)
    assert torch.allclose(z.grad.double(), zr.grad, rtol=tol, atol=tol)
    # column sums of the f64 reference gradient (the kernel sums its f32 values before they are rounded to bf16)
    want = b0.double() + zr.grad.sum(0)
    assert torch.allclose(bias.grad.double(), want, rtol=1e-4 if dt == torch.float32 else 5e-3,
                          atol=2e-3 if dt == torch.float32 else 0.3)


def test_grouped_parameter_gradient_entry_points(device):
    """mbv_wgrad_small_f32_group / mbv_colsum_accum_group through the C ABI: several products / column sums of different
    shapes, dtypes and row strides in one call each, against f64 torch expressions; accumulation into non-zero
    destinations; a bias entry that is NULL."""
    import ctypes
    from mask_bev_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(400, 256, 256), (400, 2048, 256), (37, 48, 40), (400, 256, 2048), (2, 32, 32)]
    gs = [torch.randn(t, o, generator=g).to(device) for t, o, i in shapes]
    xs = [torch.randn(t, i, generator=g).to(device) for t, o, i in shapes]
    accs = [torch.randn(o, i, generator=g).to(device) for t, o, i in shapes]
    bias = [torch.randn(o, generator=g).to(device) if j != 2 else None for j, (t, o, i) in enumerate(shapes)]
    want_w = [a.double() + gg.double().t() @ xx.double() for a, gg, xx in zip(accs, gs, xs)]
    want_b = [None if b is None else b.double() + gg.double().sum(0) for b, gg in zip(bias, gs)]
    n = len(shapes)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
    rc = lib.mbv_wgrad_small_f32_group(PA(*[t.data_ptr() for t in gs]), PA(*[t.data_ptr() for t in xs]),
                                       PA(*[t.data_ptr() for t in accs]),
                                       PA(*[(b.data_ptr() if b is not None else 0) for b in bias]),
                                       IA(*[s[0] for s in shapes]), IA(*[s[1] for s in shapes]),
                                       IA(*[s[2] for s in shapes]), n, st)
    assert rc == 0
    for a, w in zip(accs, want_w):
        assert torch.allclose(a.double(), w, rtol=1e-5, atol=2e-5 * 400 ** 0.5)
    for b, w in zip(bias, want_b):
        if b is not None:
            assert torch.allclose(b.double(), w, rtol=1e-5, atol=2e-5 * 400 ** 0.5)
    # column sums: f32 / bf16 / fp16 blocks, one of them a strided window of a wider matrix (LayerNorm partial rows)
    mats = [torch.randn(3000, 96, generator=g).to(device), torch.randn(65536, 384, generator=g).to(device).bfloat16(),
            torch.randn(777, 132, generator=g).to(device).half(), torch.randn(500, 3 * 192, generator=g).to(device)]
    views = [(mats[0], 0, 96, 96), (mats[1], 0, 384, 384), (mats[2], 0, 132, 132), (mats[3], 192, 192, 3 * 192)]
    outs = [torch.randn(v[2], generator=g).to(device) for v in views]
    want = [o.double() + m.double()[:, off:off + nn].sum(0) for o, (m, off, nn, ld) in zip(outs, views)]
    n = len(views)
    PA, IA, LA = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
    flag = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
    rc = lib.mbv_colsum_accum_group(PA(*[m.data_ptr() + off * m.element_size() for m, off, nn, ld in views]),
                                    IA(*[flag[m.dtype] for m, *_ in views]), LA(*[m.shape[0] for m, *_ in views]),
                                    IA(*[nn for _, _, nn, _ in views]), LA(*[ld for *_, ld in views]),
                                    PA(*[o.data_ptr() for o in outs]), n, st)
    assert rc == 0
    for o, w, (m, *_rest) in zip(outs, want, views):
        assert torch.allclose(o.double(), w, rtol=1e-5, atol=1e-3 * (m.shape[0] ** 0.5) * 1e-2 + 1e-4)


def test_grouped_gradient_work_equals_per_layer_launches(device, monkeypatch):
    """The end-of-backward grouping of the small parameter-gradient work (ops.flush_deferred_grads) is a reordering of
    launches only: a flattened module gives the same arena gradient with MBV_WGRAD_GROUP=0 (per-layer launches) and with
    the grouping on — same kernels' arithmetic, f32 atomics in another order.  fp32 compute, tolerance 5e-3 of the
    largest entry per parameter: what two runs of ONE mode differ by (the backward's f32 atomics are unordered;
    test_arena_gradients_equal_plain_autograd uses the same bound).  A backward pass that raises half-way must not leak
    its pending work into the next pass."""
    from mask_bev_amd import ops
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = dict(tiny_kwargs(nx=96, ny=96, q=8), compute_dtype='fp32')
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    batch = (scans, (labels.to(device), gt.to(device)))
    grads = {}
    for mode in ('0', '1'):
        switches.patch(monkeypatch, wgrad_group=mode)
        switches.patch(monkeypatch, tn_group=mode)
        torch.manual_seed(0)
        m = MaskBevModule(**kw).to(device).train()
        m.log_scalars = False
        h = m._panoptic_head._panoptic_head
        h.num_points, h.point_seed = 600, 5
        arena = m.flatten_parameters()
        if mode == '1':
            # a pass that dies after some work was queued: nothing of it may reach the arena later
            class Boom(torch.autograd.Function):
                @staticmethod
                def forward(ctx, x):
                    return x.clone()

                @staticmethod
                def backward(ctx, g):
                    raise RuntimeError('boom')
            lin = m._panoptic_head._panoptic_head.cls_embed
            x = torch.randn(16, lin.in_features, device=device, requires_grad=True)
            with pytest.raises(RuntimeError, match='boom'):
                lin(Boom.apply(x)).sum().backward()
            torch.cuda.synchronize()
            arena.zero_grad()
        m.training_step(batch, 0).backward()
        torch.cuda.synchronize()
        assert not ops._PENDING or all(not any(lists) for lists in ops._PENDING.values()) or mode == '1'
        grads[mode] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    ops._PENDING.clear()
    for n, g0 in grads['0'].items():
        g1 = grads['1'][n]
        scale = float(g0.abs().max()) + 1e-12
        assert float((g0 - g1).abs().max()) <= 5e-3 * scale + 1e-9, n


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_layernorm_affine_update_inside_k3_backward_is_bit_identical(device, dtype):
    """Round 6 (mbv_scatter_layernorm_bwd_adamw, FlatAdam.fuse_layernorm_affine): K3's backward performs the AdamW update
    of the two (C, ny, nx) LayerNorm affine parameters itself — their gradients never reach the arena.  Against the plain
    form (K3 backward accumulates into the arena gradient, mbv_adamw_step updates everything) from the same initial state,
    on the same scans and upstream gradients, three steps: parameters, both moments and the 16-bit shadow BIT FOR BIT
    (one definition of the update arithmetic, csrc/adam.hpp), the gradient range of the two parameters stays zero, a second
    backward before step() raises, and a data-parallel gradient scale leaves the fusion unused."""
    from mask_bev_amd import ops
    from mask_bev_amd._lib import MaskBevHipError
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = dict(tiny_kwargs(nx=96, ny=96, q=8), compute_dtype=dtype)
    scans = [[x.to(device) for x in random_scans(kw, [2500, 3000], seed=s)] for s in range(3)]

    def run(fused: bool):
        torch.manual_seed(3)
        m = MaskBevModule(**kw).to(device).train()
        arena = m.flatten_parameters()
        opt = m.configure_optimizers()['optimizer']
        ln = m._encoder._layer_norm
        assert opt.fuse_layernorm_affine(ln.weight if fused else None, ln.bias if fused else None) == fused
        gen = torch.Generator(device='cpu').manual_seed(11)
        for s in range(3):
            with m._autocast():
                x = m._encoder(scans[s], patch=m._patch_handoff())
            rows = x.rows if isinstance(x, ops.PatchTokens) else x
            up = (torch.randn(rows.shape, generator=gen) * 0.1).to(device=device, dtype=rows.dtype)
            rows.backward(up)
            if fused:
                lo, hi = arena.range_of(ln)
                assert float(arena.grad[lo:hi].abs().max()) == 0.0       # the gradient never reached the arena
                if s == 0:
                    with m._autocast():
                        y = m._encoder(scans[s], patch=m._patch_handoff())
                    yr = y.rows if isinstance(y, ops.PatchTokens) else y
                    with pytest.raises(MaskBevHipError, match='twice before step'):
                        yr.backward(up)
            opt.step()
        torch.cuda.synchronize()
        out = (arena.param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
               None if arena.shadow is None else arena.shadow.clone())
        opt.fuse_layernorm_affine(None)
        assert ops.K3_ADAM[0] is None
        return out

    plain, fused = run(False), run(True)
    for a, b, name in zip(plain, fused, ('param', 'exp_avg', 'exp_avg_sq', 'shadow')):
        if a is not None:
            assert torch.equal(a, b), f'{name}: max |diff| {float((a.float() - b.float()).abs().max()):.3e}'
    # grad_scale != 1 (a data-parallel step): the claim is refused, the plain path runs
    torch.manual_seed(3)
    m = MaskBevModule(**kw).to(device).train()
    arena = m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    ln = m._encoder._layer_norm
    assert opt.fuse_layernorm_affine(ln.weight, ln.bias)
    opt.grad_scale = 0.5
    with m._autocast():
        x = m._encoder(scans[0], patch=m._patch_handoff())
    rows = x.rows if isinstance(x, ops.PatchTokens) else x
    rows.backward(torch.ones_like(rows))
    lo, hi = arena.range_of(ln)
    assert float(arena.grad[lo:hi].abs().max()) > 0.0
    opt.step()
    opt.fuse_layernorm_affine(None)
