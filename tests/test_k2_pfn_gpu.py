"""K2 PillarFeatureNet (K2a decoration + K2b per-pillar kernels, padded-row algebra) vs the oracle's dense
zero-padded PFN: output, gradients of every parameter, BatchNorm running statistics; train and eval mode.
f32: output rtol 1e-4, gradients 2e-3 in the L2 norm and 1e-2 of the tensor's max (gate flips, see the test)."""
import pytest
import torch

from oracle import maskbev_oracle as O
from tests.util_cfg import random_scans

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


@pytest.mark.parametrize('pc_dim', [4, 3])
@pytest.mark.parametrize('chans,P,sizes,training', [([32, 32, 32], 8, [3000, 2000], True),
                                                    ([128, 128, 128], 32, [6000], True),
                                                    ([64, 128], 4, [1500, 10, 900], True),
                                                    ([32, 32, 32], 8, [2500], False),
                                                    # unit counts outside {32, 64, 128}: the lane = channel walk kernels
                                                    # (the 4-channel lane map takes the three sizes above) and library Linears
                                                    ([48, 96], 8, [2000, 700], True)])
@pytest.mark.parametrize('path', ['stream', 'walk', 'layers'])
def test_pfn_matches_dense_oracle(device, chans, P, sizes, training, pc_dim, path, monkeypatch):
    """path: 'stream' = the one-call forward with the pillar term inside K2c and streamed BatchNorm statistics (the default),
    'walk' = the one-call forward with the per-pillar statistics walk, 'layers' = one C-ABI call per kernel."""
    from mask_bev_amd import ops, switches
    from mask_bev_amd.encoders import PillarFeatureNet
    switches.patch(monkeypatch, pfn_stream_stats=(path == 'stream'), pfn_one_call=(path != 'layers'))
    kw = dict(x_range=(-10, 10), y_range=(-10, 10), z_range=(-3, 1), voxel_size=0.25, num_queries=4, max_num_points=P,
              encoder_feat_channels=chans, backbone_embed_dim=48, head_feat_channels=128, head_out_channels=128,
              pc_point_dim=pc_dim)      # 3 = xyz only (Waymo, mask_bev_module.py:74): a 10-channel decoration
    cfg = O.make_cfg(**kw)
    sd = {k: v for k, v in O.make_state_dict(cfg, 3).items() if k.startswith(O.ENC + '_voxel_encoder')}
    for k in list(sd):
        if k.endswith('running_mean'):
            sd[k] = torch.randn_like(sd[k]) * 0.1
        if k.endswith('running_var'):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    scans = random_scans(kw, sizes, seed=P)
    # oracle (dense, zero padded)
    voxels, nump, coors = O.voxelize(cfg, scans)
    sd_g = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running_' not in k else v.clone())
            for k, v in sd.items()}
    bufs = {k: v.clone() for k, v in sd.items() if 'running_' in k}
    ref = O.pfn_forward(cfg, sd_g, voxels, nump, coors, training, bn_buffers=bufs)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    ref.backward(go)
    # product
    net = PillarFeatureNet(in_channels=pc_dim, feat_channels=chans, with_distance=True, voxel_size=cfg.voxel_size3,
                           point_cloud_range=cfg.pc_range)
    net.load_state_dict({k[len(O.ENC + '_voxel_encoder.'):]: v for k, v in sd.items()})
    net = net.to(device).train(training)
    geom = ops.VoxelGeometry.from_ranges(cfg.pc_range, cfg.voxel_size3)
    pil = ops.voxelize([s.to(device) for s in scans], geom, P, cfg.max_voxels)
    out = net(pil)
    out.backward(go.to(device))
    assert out.shape == ref.shape
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-4
    for name, prm in net.named_parameters():
        r = sd_g[O.ENC + '_voxel_encoder.' + name].grad
        assert r is not None, name
        g = prm.grad.cpu()
        # L2-relative 2e-3; in the max norm 1e-2: a ReLU gate / max-over-points winner whose pre-activations differ in
        # the last bit between the dense and the per-pillar evaluation order moves ONE row's contribution (|g x| up to
        # a few units of a (U, 10) weight gradient whose largest entries are ~30), seen as 2.7e-3 of the max with xyz-only
        # points and random running statistics — a discontinuity of the function, not an accumulation error
        assert float((g - r).norm() / r.norm().clamp(min=1e-12)) < 2e-3, name
        assert _rel(g, r) < 1e-2, name
    for name, buf in net.named_buffers():
        if 'running_' in name:
            want = bufs[O.ENC + '_voxel_encoder.' + name]
            torch.testing.assert_close(buf.cpu(), want, rtol=1e-4, atol=1e-6)
        if 'num_batches_tracked' in name:
            assert int(buf) == (1 if training else 0)


@pytest.mark.gpu
@pytest.mark.parametrize('group', [1, 2])
def test_fourier_encoder_matches_dense_oracle(device, group):
    """A3 (mask_bev_encoders.py:85-89): `encoder_encoding_type='fourier'` — the product evaluates the per-point
    Fourier MLP and the PFN on real points only (padded slots enter through the (P - n) e0 term of points_mean);
    the oracle encodes the dense zero-padded (V, P, 4) tensor like the reference.  Pseudo-image within 1e-4, and
    the gradients of the encoding's own parameters and of the first PFN layer within 1e-3."""
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from oracle import maskbev_oracle as O
    from tests.util_cfg import random_scans, tiny_kwargs
    kw = dict(tiny_kwargs(nx=48, ny=48, p=8), encoder_encoding_type='fourier', encoder_fourier_enc_group=group)
    cfg = O.make_cfg(**kw)
    sd = O.make_state_dict(cfg, 3)
    m = MaskBevModule(**kw)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).train()
    scans = random_scans(kw, [1500, 900], seed=5)
    out = m.forward_encode([s.to(device) for s in scans])
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    (out * w.to(device)).sum().backward()
    sd_g = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running_' not in k else v.clone())
            for k, v in sd.items()}
    ref = O.encoder_forward(cfg, sd_g, scans, training=True)
    (ref * w).sum().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 1e-4
    got = dict(m.named_parameters())
    for k in ['_encoder._pos_encoder.Wr.weight', '_encoder._pos_encoder.mlp.0.weight', '_encoder._pos_encoder.mlp.2.bias',
              '_encoder._voxel_encoder.pfn_layers.0.linear.weight', '_encoder._voxel_encoder.pfn_layers.1.norm.weight']:
        g, r = got[k].grad.cpu(), sd_g[k].grad
        assert float((g - r).abs().max() / r.abs().max().clamp(min=1e-9)) < 1e-3, k


@pytest.mark.parametrize('m,c,n,ldw,nk', [(440668 // 8, 64, 128, 128, True), (5000, 10, 64, 10, True), (5000, 11, 64, 11, True),
                                         (33, 64, 64, 128, True), (4097, 128, 64, 128, False), (9000, 64, 64, 128, False),
                                         (9000, 128, 128, 128, False), (70, 16, 32, 16, True), (1000, 64, 96, 64, True)])
def test_skinny_gemm_f32_equals_float64_product(device, m, c, n, ldw, nk):
    """K2c (mbv_skinny_gemm_f32): x (m, c) . w^T / x . w in exact f32 on MFMA for every register-tile instantiation, row tails,
    contraction lengths that are not multiples of 4, and a weight that is a column block of a wider matrix."""
    from mask_bev_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + c + n)
    x = torch.randn(m, c, generator=g).to(device)
    if nk:
        wfull = torch.randn(n, ldw, generator=g).to(device)
        w = wfull[:, ldw - c:]                                   # the LAST c columns: a strided view with an offset
        ref = x.double() @ w.double().t()
    else:
        wfull = torch.randn(c, ldw, generator=g).to(device)
        w = wfull[:, :n]
        ref = x.double() @ w.double()
    assert lib.mbv_skinny_gemm_f32_supported(m, c, n)
    y = torch.full((m, n), float('nan'), device=device)
    ops.check(lib.mbv_skinny_gemm_f32(ops._ptr(x), ops._ptr(w), ops._ptr(y), m, c, n, int(w.stride(0)), 1 if nk else 0,
                                      ops._stream()), 'mbv_skinny_gemm_f32')
    assert torch.isfinite(y).all()
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * max(1.0, c ** 0.5)
