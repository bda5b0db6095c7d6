"""GPU parity of K1 (voxelise), K2a (decorate) and K3 (scatter + CHW LayerNorm) against the oracle.
Integer outputs must be bit-exact; f32 outputs within the tolerances written below."""
import numpy as np
import pytest
import torch

from oracle import maskbev_oracle as O

pytestmark = pytest.mark.gpu


def _cfg(nx=64, ny=64, vs=0.25, P=8, C=32, pc_dim=4, max_voxels=250000):
    return O.make_cfg(x_range=(-nx * vs / 2, nx * vs / 2), y_range=(-ny * vs / 2, ny * vs / 2), z_range=(-3, 1),
                      voxel_size=vs, num_queries=8, max_num_points=P, encoder_feat_channels=[C, C, C],
                      backbone_embed_dim=24, head_feat_channels=32, head_out_channels=32, pc_point_dim=pc_dim,
                      max_voxels=max_voxels)


def _geom(cfg):
    from mask_bev_amd.ops import VoxelGeometry
    return VoxelGeometry.from_ranges(cfg.pc_range, cfg.voxel_size3)


def _scans(cfg, sizes, seed, dim=4, spread=1.2):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in sizes:
        p = torch.rand(n, dim, generator=g)
        lo = torch.tensor([cfg.x_range[0], cfg.y_range[0], cfg.z_range[0]])
        hi = torch.tensor([cfg.x_range[1], cfg.y_range[1], cfg.z_range[1]])
        mid, half = (lo + hi) / 2, (hi - lo) / 2 * spread
        p[:, :3] = mid + (p[:, :3] * 2 - 1) * half
        out.append(p)
    return out


def _check_voxelize(cfg, scans, device):
    from mask_bev_amd import ops
    pil = ops.voxelize([s.to(device) for s in scans], _geom(cfg), cfg.max_num_points, cfg.max_voxels)
    voxels = ops.gather_voxels(pil)
    torch.cuda.synchronize()
    ref_v, ref_n, ref_c = O.voxelize(cfg, scans)
    assert pil.num_pillars == ref_c.shape[0]
    assert torch.equal(pil.coors.cpu(), ref_c.to(torch.int32))
    assert torch.equal(pil.num_points.cpu(), ref_n.to(torch.int32))
    assert torch.equal(voxels.cpu(), ref_v)                       # bit-exact copy of the kept points
    assert pil.num_rows == int(ref_n.sum())
    return pil, (ref_v, ref_n, ref_c)


@pytest.mark.parametrize('sizes', [[3000], [5000, 1, 2500], [0, 700], [64, 64, 64, 64]])
def test_voxelize_bit_exact_random(device, sizes):
    cfg = _cfg()
    _check_voxelize(cfg, _scans(cfg, sizes, seed=sum(sizes)), device)


def test_voxelize_crowded_cells_and_order(device):
    """> P points per cell, all points in one cell, and duplicate cells in different order."""
    cfg = _cfg(nx=8, ny=8, vs=1.0, P=4)
    g = torch.Generator().manual_seed(5)
    a = torch.rand(4000, 4, generator=g) * torch.tensor([8.0, 8.0, 4.0, 1.0]) - torch.tensor([4.0, 4.0, 3.0, 0.0])
    one = torch.rand(3000, 4, generator=g) * torch.tensor([0.5, 0.5, 1.0, 1.0])      # a single pillar
    _check_voxelize(cfg, [a, one, a.flip(0)], device)


def test_voxelize_borders(device):
    """Points exactly on cell borders and on the range bounds (strict < pre-filter)."""
    cfg = _cfg(nx=16, ny=16, vs=0.5, P=3)
    xs = torch.arange(-4.0, 4.0001, 0.25)
    pts = torch.stack(torch.meshgrid(xs, xs, torch.tensor([-3.0, -2.9999, 0.0, 0.9999, 1.0]), indexing='ij'), -1)
    pts = torch.cat([pts.reshape(-1, 3), torch.zeros(pts.numel() // 3, 1)], 1)
    pts = pts[torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(1))]
    eps = torch.tensor([[np.nextafter(np.float32(4.0), np.float32(0.0)), 0.0, 0.0, 0.0],
                        [np.nextafter(np.float32(-4.0), np.float32(0.0)), 0.0, 0.0, 0.0],
                        [float('nan'), 0.0, 0.0, 0.0], [float('inf'), 0.0, 0.0, 0.0]])
    _check_voxelize(cfg, [torch.cat([pts, eps])], device)


def test_voxelize_max_voxels_cap(device):
    cfg = _cfg(max_voxels=100)
    _check_voxelize(cfg, _scans(cfg, [4000, 50, 3000], seed=3), device)


def test_voxelize_lidar_sized(device):
    """Full-size 120k-point scans on the 512x512 grid, against the C oracle."""
    cfg = O.make_cfg(x_range=(-40, 40), y_range=(-40, 40), z_range=(-3, 1), voxel_size=0.15625, num_queries=100,
                     max_num_points=32, encoder_feat_channels=[128, 128, 128], backbone_embed_dim=192,
                     head_feat_channels=256, head_out_channels=256)
    assert (cfg.nx, cfg.ny) == (512, 512)
    g = torch.Generator().manual_seed(11)
    scans = []
    for _ in range(2):
        r = torch.rand(120000, generator=g) ** 2 * 55
        th = torch.rand(120000, generator=g) * 6.2831853
        z = torch.rand(120000, generator=g) * 5 - 3.5
        scans.append(torch.stack([r * th.cos(), r * th.sin(), z, torch.rand(120000, generator=g)], 1))
    pil, _ = _check_voxelize(cfg, scans, device)
    # size-independent properties
    c2p = pil.cell_to_pillar.cpu()
    coors = pil.coors.cpu().long()
    assert torch.equal(c2p[coors[:, 0], coors[:, 2] * cfg.nx + coors[:, 3]], torch.arange(pil.num_pillars, dtype=torch.int32))
    assert int((c2p >= 0).sum()) == pil.num_pillars
    rs = pil.row_start.cpu()
    assert torch.equal(rs[1:] - rs[:-1], pil.num_points.cpu())


def test_decorate_matches_oracle(device):
    from mask_bev_amd import ops
    cfg = _cfg()
    scans = _scans(cfg, [4000, 3000], seed=9)
    pil, (rv, rn, rc) = _check_voxelize(cfg, scans, device)
    rows, row_pillar = ops.pfn_decorate(pil, cfg.voxel_size3, cfg.pc_range)
    dense = O.pfn_decorate(cfg, rv, rn, rc)                              # (V, P, 11), zero padded
    mask = torch.arange(cfg.max_num_points).view(1, -1) < rn.view(-1, 1)
    ref_rows = dense[mask]                                               # pillar-major, slot order == compact order
    torch.testing.assert_close(rows.cpu(), ref_rows, rtol=1e-5, atol=1e-5)
    ref_pillar = torch.arange(rv.shape[0]).view(-1, 1).expand(-1, cfg.max_num_points)[mask]
    assert torch.equal(row_pillar.cpu(), ref_pillar)


@pytest.mark.parametrize('nx,ny,C,sizes', [(64, 64, 32, [3000, 2000]), (52, 40, 16, [500]), (63, 32, 8, [900, 10, 0])])
def test_scatter_layernorm_fwd_bwd(device, nx, ny, C, sizes):
    """K3 vs dense scatter + F.layer_norm (f32): forward and all three gradients, rtol 1e-4 / atol 1e-5."""
    from mask_bev_amd import ops
    cfg = _cfg(nx=nx, ny=ny, C=C)
    scans = _scans(cfg, sizes, seed=nx)
    pil, (rv, rn, rc) = _check_voxelize(cfg, scans, device)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(pil.num_pillars, C, generator=g)
    w = 1 + 0.1 * torch.randn(C, ny, nx, generator=g)
    b = 0.1 * torch.randn(C, ny, nx, generator=g)
    go = torch.randn(len(scans), C, ny, nx, generator=g)
    # oracle
    f_ref, w_ref, b_ref = feats.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    canvas = O.scatter_to_canvas(cfg, f_ref, rc, len(scans))
    out_ref = torch.nn.functional.layer_norm(canvas, [C, ny, nx], w_ref, b_ref, 1e-3)
    out_ref.backward(go)
    # product
    f_d, w_d, b_d = (t.clone().to(device).requires_grad_() for t in (feats, w, b))
    out = ops.scatter_layernorm(f_d, w_d, b_d, pil, len(scans), ny, nx, 1e-3)
    out.backward(go.to(device))
    torch.testing.assert_close(out.detach().cpu(), out_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(w_d.grad.cpu(), w_ref.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b_d.grad.cpu(), b_ref.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(f_d.grad.cpu(), f_ref.grad, rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('owned', [False, True])
def test_scatter_layernorm_leaves_the_maps_absmax_record(device, owned):
    """fp32 compute: K3 leaves the absmax record of its f32 map (one atomic per workgroup) as the hint the K20 patch
    projection reads — a fresh record for a fresh map, ONE persistent record (cleared and rewritten every call) for a
    caller-owned buffer that a captured graph reads."""
    from mask_bev_amd import ops, switches
    cfg = _cfg(nx=64, ny=64, C=32)
    scans = _scans(cfg, [3000, 2000], seed=9)
    pil, _ = _check_voxelize(cfg, scans, device)
    g = torch.Generator().manual_seed(3)
    w = (1 + 0.1 * torch.randn(32, 64, 64, generator=g)).to(device)
    b = (0.1 * torch.randn(32, 64, 64, generator=g)).to(device)
    buf = torch.empty((2, 32, 64, 64), device=device) if owned else None
    if owned:
        ops.static_amax_register(buf)          # (what graph.py does for the static input of its captured step)
    recs = []
    with switches.override(amax_hints=True, ln_bound_hints=True, gemm32s=True):
        for scale in (1.0, 50.0):
            feats = (torch.randn(pil.num_pillars, 32, generator=g) * scale).to(device)
            out = ops.scatter_layernorm(feats, w * scale, b, pil, 2, 64, 64, 1e-3, out=None if buf is None else buf.detach())
            rec = ops.amax_hint_get(out)
            assert rec is not None
            assert int(rec.max()) == int(out.abs().max().view(torch.int32))
            recs.append(rec)
    assert (recs[0].data_ptr() == recs[1].data_ptr()) == owned


@pytest.mark.gpu
@pytest.mark.parametrize('nx,ny,C,sizes', [(64, 64, 32, [3000, 2000]), (52, 40, 64, [500]), (512, 8, 32, [900, 10, 0]),
                                           (300, 12, 32, [2000, 1500])])
@pytest.mark.parametrize('lo', [torch.bfloat16, torch.float16])
def test_scatter_layernorm_patch_tokens(device, nx, ny, C, sizes, lo):
    """K3 patch-token layout (bf16 rows of the backbone's 4 x 4 patch projection) vs the oracle's dense scatter +
    F.layer_norm: the forward values are the bf16 rounding of the f32 map (<= 1 bf16 ulp of the f32 reference), and
    a gradient arriving in the same layout gives the gradients of the NCHW path (rtol 1e-4 / atol 2e-5)."""
    from mask_bev_amd import ops
    cfg = _cfg(nx=nx, ny=ny, C=C)
    scans = _scans(cfg, sizes, seed=nx + 1)
    pil, (rv, rn, rc) = _check_voxelize(cfg, scans, device)
    B = len(scans)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(pil.num_pillars, C, generator=g)
    w = 1 + 0.1 * torch.randn(C, ny, nx, generator=g)
    b = 0.1 * torch.randn(C, ny, nx, generator=g)
    go = torch.randn(B, C, ny, nx, generator=g).to(lo)                         # what the 16-bit dgrad GEMM would hand back
    f_ref, w_ref, b_ref = feats.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    canvas = O.scatter_to_canvas(cfg, f_ref, rc, B)
    out_ref = torch.nn.functional.layer_norm(canvas, [C, ny, nx], w_ref, b_ref, 1e-3)
    out_ref.backward(go.float())
    f_d, w_d, b_d = (t.clone().to(device).requires_grad_() for t in (feats, w, b))
    assert ops.patch_layout_supported(C, ny, nx, 4)
    with torch.autocast('cuda', dtype=lo):               # the patch rows take the 16-bit type of the autocast region
        tok = ops.scatter_layernorm(f_d, w_d, b_d, pil, B, ny, nx, 1e-3, patch=4)
    assert isinstance(tok, ops.PatchTokens) and tok.rows.dtype == lo
    assert tuple(tok.rows.shape) == (B, ny // 4, nx // 4, 16 * C)
    # forward: identical to rounding the f32 NCHW result of the same kernel family, and within bf16 of the oracle
    with torch.no_grad():
        img32 = ops.scatter_layernorm(f_d, w_d, b_d, pil, B, ny, nx, 1e-3)
    assert torch.equal(tok.to_image(), img32.to(lo))
    torch.testing.assert_close(tok.to_image().float().cpu(), out_ref.detach(),
                               rtol=8e-3 if lo == torch.bfloat16 else 1e-3, atol=1e-5)
    # backward: the gradient in patch layout, element (y%4)*4C + c*4 + x%4 of row (b, y/4, x/4)
    go_rows = go.view(B, C, ny // 4, 4, nx // 4, 4).permute(0, 2, 4, 3, 1, 5).reshape(B, ny // 4, nx // 4, 16 * C)
    tok.rows.backward(go_rows.contiguous().to(device))
    torch.testing.assert_close(w_d.grad.cpu(), w_ref.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b_d.grad.cpu(), b_ref.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(f_d.grad.cpu(), f_ref.grad, rtol=1e-4, atol=2e-5)
    # destination buffer variant writes the same bytes
    buf = torch.zeros_like(tok.rows)
    tok2 = ops.scatter_layernorm(f_d.detach(), w_d.detach(), b_d.detach(), pil, B, ny, nx, 1e-3, patch=4, out=buf)
    assert tok2.rows.data_ptr() == buf.data_ptr() and torch.equal(buf, tok.rows)


@pytest.mark.gpu
def test_scatter_layernorm_patch_unsupported_shapes(device):
    from mask_bev_amd import ops
    assert not ops.patch_layout_supported(32, 62, 64, 4)       # ny not a multiple of the patch
    assert not ops.patch_layout_supported(24, 64, 64, 4)       # partial channel tile
    assert not ops.patch_layout_supported(32, 64, 64, 2)       # only the 4 x 4 projection of the MaskBEV backbone


@pytest.mark.gpu
def test_patch_embed_tokens_equal_conv(device):
    """PatchEmbed fed K3's patch rows (one GEMM) vs the same values through the reference's Conv2d route
    (swin.py:579-586): outputs and all parameter / input gradients agree to bf16 accuracy."""
    from mask_bev_amd import ops
    from mask_bev_amd.layers import PatchEmbed
    torch.manual_seed(3)
    B, C, ny, nx, E = 2, 32, 32, 48, 96
    pe = PatchEmbed(C, E, 4).to(device)
    img = torch.randn(B, C, ny, nx, device=device).bfloat16()
    rows = img.view(B, C, ny // 4, 4, nx // 4, 4).permute(0, 2, 4, 3, 1, 5).reshape(B, ny // 4, nx // 4, 16 * C)
    rows = rows.contiguous().requires_grad_()
    tok = ops.PatchTokens(rows, C, 4)
    assert torch.equal(tok.to_image(), img)
    go = torch.randn(B, ny // 4, nx // 4, E, device=device)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y_tok = pe(tok)
    y_tok.float().backward(go)
    grads_tok = [p.grad.clone() for p in pe.parameters()]
    g_rows = rows.grad.clone()
    for p in pe.parameters():
        p.grad = None
    img_r = img.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y_conv = pe(img_r)
    y_conv.float().backward(go)
    torch.testing.assert_close(y_tok.float(), y_conv.float(), rtol=2e-2, atol=2e-2)
    for (n, p), gt in zip(pe.named_parameters(), grads_tok):
        torch.testing.assert_close(gt, p.grad, rtol=2e-2, atol=2e-2 * float(p.grad.abs().max()), msg=lambda m: f'{n}: {m}')
    g_img = ops.PatchTokens(g_rows, C, 4).to_image()
    torch.testing.assert_close(g_img.float(), img_r.grad.float(), rtol=2e-2, atol=2e-2 * float(img_r.grad.abs().max()))


@pytest.mark.gpu
def test_scatter_layernorm_full_size_properties(device):
    """K3 at the bench size (B 4, C 128, 512 x 512, 120 k-point scans): size-independent properties instead of a dense
    oracle — per-scan statistics of the un-affined output, the closed form of every empty cell, and agreement of
    the bf16 patch rows with the f32 image."""
    from mask_bev_amd import ops
    cfg = _cfg(nx=512, ny=512, vs=0.15625, P=32, C=128)
    scans = _scans(cfg, [120000, 120000, 90000, 120000], seed=3, spread=1.0)
    pil = ops.voxelize([s.to(device) for s in scans], _geom(cfg), cfg.max_num_points, cfg.max_voxels)
    B, C, ny, nx = 4, 128, 512, 512
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(pil.num_pillars, C, generator=g).to(device)
    w = torch.ones(C, ny, nx, device=device)
    b = torch.zeros(C, ny, nx, device=device)
    out = ops.scatter_layernorm(feats, w, b, pil, B, ny, nx, 1e-3)                     # identity affine: x-hat itself
    flat = out.view(B, -1).double()
    assert float(flat.mean(1).abs().max()) < 1e-4
    # eps = 1e-3 inside the rsqrt: var(out) = v / (v + eps) with v the variance of the scattered canvas, which is
    # known from the pillar rows alone (every empty cell is 0)
    starts = pil.pillar_batch_start.cpu().tolist()
    for s in range(B):
        rows = feats[starts[s]:starts[s + 1]].double()
        n_el = C * ny * nx
        mean = float(rows.sum()) / n_el
        v = float((rows * rows).sum()) / n_el - mean * mean
        assert abs(float(flat[s].var(unbiased=False)) - v / (v + 1e-3)) < 1e-5
    # every empty cell holds the same value per scan: (0 - mean) * rstd
    occ = (pil.cell_to_pillar.view(B, ny, nx) >= 0)
    for s in range(B):
        empty = out[s][:, ~occ[s]]
        assert float(empty.max() - empty.min()) == 0.0
    # affine + patch rows: equal to the bf16 rounding of the f32 image of the same kernel family
    w2 = 1 + 0.1 * torch.randn(C, ny, nx, generator=g).to(device)
    b2 = 0.1 * torch.randn(C, ny, nx, generator=g).to(device)
    img = ops.scatter_layernorm(feats, w2, b2, pil, B, ny, nx, 1e-3)
    tok = ops.scatter_layernorm(feats, w2, b2, pil, B, ny, nx, 1e-3, patch=4)
    assert torch.equal(tok.to_image(), img.bfloat16())
    torch.testing.assert_close(img, out * w2 + b2, rtol=1e-5, atol=1e-5)
