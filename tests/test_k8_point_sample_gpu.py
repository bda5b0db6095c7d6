"""K8 indexed bilinear point sampling: forward/backward vs F.grid_sample on gathered maps (f32, rtol 1e-5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,H,W,G,P,shared', [(12, 16, 16, 7, 300, False), (8, 128, 128, 8, 2000, True),
                                              (5, 33, 21, 5, 64, False), (3, 200, 160, 3, 500, False),
                                              # maps larger than one LDS tile (round 6): row bands, forward (dense sampling)
                                              # and backward; a sampled subset (zero fill + band stores); ragged last band;
                                              # a width that leaves the bands' first pixel unaligned
                                              (4, 256, 256, 3, 9000, False), (3, 200, 160, 3, 4100, False),
                                              (2, 130, 250, 2, 4100, True), (2, 256, 256, 2, 12544, True)])
def test_point_sample_fwd_bwd(device, N, H, W, G, P, shared):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(N * 7 + P)
    src = torch.randn(N, H, W, generator=g)
    src_index = torch.randperm(N, generator=g)[:G]
    n_coord = 2 if shared else G
    coords = torch.rand(n_coord, P, 2, generator=g) * 1.2 - 0.1            # some points outside → zero padding
    coord_index = (torch.arange(G) % n_coord)
    go = torch.randn(G, P, generator=g)
    s_r = src.clone().requires_grad_()
    ref = F.grid_sample(s_r[src_index].unsqueeze(1), 2.0 * coords[coord_index].unsqueeze(2) - 1.0,
                        align_corners=False).squeeze(3).squeeze(1)
    ref.backward(go)
    s_d = src.clone().to(device).requires_grad_()
    out = ops.point_sample(s_d, src_index.to(device=device, dtype=torch.int32), coords.to(device),
                           coord_index.to(device=device, dtype=torch.int32))
    out.backward(go.to(device))
    # pixel coordinates of magnitude W carry ~ W * 2^-24 of f32 rounding: the weight error bound against torch's arithmetic
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=max(1e-5, W * 2e-7))
    torch.testing.assert_close(s_d.grad.cpu(), s_r.grad, rtol=1e-4, atol=max(1e-5, W * 2e-7))


@pytest.mark.parametrize('H,W', [(256, 256), (200, 160), (130, 250)])
def test_banded_forward_is_bit_identical_to_the_gather_form(device, H, W):
    """A map larger than one LDS tile, sampled densely, goes through row bands in LDS (k_point_sample_fwd_bands); sampled
    sparsely (few points) through the per-point gather kernel.  Same bilinear set-up, same sum order: the values of the same
    points must be EQUAL, including points whose taps straddle two bands and points outside the map."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(H + W)
    src = torch.randn(3, H, W, generator=g).to(device)
    p = H * W // 8 + 64
    coords = (torch.rand(3, p, 2, generator=g) * 1.1 - 0.05).to(device)
    br0 = 35840 // W - 1
    nb = -(-H // br0)
    br = -(-H // nb)                                     # rows per band (csrc/point_sample.hip: equal bands of <= 140 KB)
    coords[:, :64, 1] = ((torch.arange(64, device=device) % 8 + 1) * br + 0.5 + torch.rand(3, 64, generator=g).to(device) * 0.02 - 0.01) / H   # band seams
    idx = torch.arange(3, dtype=torch.int32, device=device)
    dense = ops.point_sample(src, idx, coords, idx)
    k = 200
    sparse = ops.point_sample(src, idx, coords[:, :k].contiguous(), idx)
    assert torch.equal(dense[:, :k], sparse)


@pytest.mark.parametrize('N,H,W,G,P', [(6, 64, 64, 9, 700), (3, 512, 512, 5, 3000), (2, 37, 29, 2, 100)])
def test_point_sample_packed_binary(device, N, H, W, G, P):
    """Bit-packed binary maps: identical to grid_sample on the {0,1} float maps (duplicated indices allowed)."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(N + H)
    src = (torch.rand(N, H, W, generator=g) > 0.6).float()
    src[0] = 0                                                            # an all-zero (padding) mask
    src_index = torch.randint(0, N, (G,), generator=g)
    coords = torch.rand(G, P, 2, generator=g) * 1.1 - 0.05
    ref = F.grid_sample(src[src_index].unsqueeze(1), 2.0 * coords.unsqueeze(2) - 1.0, align_corners=False).squeeze(3).squeeze(1)
    pm = ops.pack_binary_masks(src.to(device))
    out = ops.point_sample_packed(pm, src_index.to(device=device, dtype=torch.int32), coords.to(device),
                                  torch.arange(G, dtype=torch.int32, device=device))
    # pixel coordinates of magnitude W carry ~W * 2^-24 of f32 rounding, which is the weight error bound
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=max(5e-6, W * 1.2e-7))


def test_point_sample_bf16_source_outside_autocast(device):
    """Regression: a bf16 source outside an autocast region must be converted, not reinterpreted."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(0)
    src = torch.randn(4, 32, 32, generator=g).bfloat16()
    coords = torch.rand(4, 50, 2, generator=g)
    idx = torch.arange(4, dtype=torch.int32, device=device)
    out = ops.point_sample(src.to(device), idx, coords.to(device), idx)
    ref = F.grid_sample(src.float().unsqueeze(1), 2.0 * coords.unsqueeze(2) - 1.0, align_corners=False).squeeze(3).squeeze(1)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('H,W', [(64, 64), (512, 512), (37, 29), (32, 48), (128, 24)])
def test_pack_binary_masks_bits(device, H, W):
    """bit i of word k = (pixel 32 k + i != 0), both packing kernels (whole 1024-pixel tiles / ragged)."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(H * W)
    src = (torch.rand(5, H, W, generator=g) > 0.5).float() * torch.randn(5, H, W, generator=g).sign()
    src[1] = 0
    src[2] = 1
    pm = ops.pack_binary_masks(src.to(device))
    words = pm.words.cpu().numpy().astype(np.uint32).reshape(5, -1)
    bits = (src.flatten(1).numpy() != 0)
    nw = words.shape[1]
    padded = np.zeros((5, nw * 32), dtype=bool)
    padded[:, :H * W] = bits
    want = (padded.reshape(5, nw, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    assert np.array_equal(words, want)


@pytest.mark.parametrize('hw', [(16, 24), (150, 200)])
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_stack_gradient_sink_equals_permute_and_cast(device, dt, hw):
    """mbv_point_sample_bwd_stack (ops.StackGradSink): the gradient of a (D, B, Q, H, W) stack of which every map is sampled
    once, stored sample-major (B, D, Q, H*W) in f32 / bf16 / fp16 by K8's backward itself — equal to the ordinary f32
    gradient permuted and rounded; autograd receives a zero-stride token of the stack's shape."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(17)
    d, b, q, p = 3, 2, 5, 64
    h, w = hw                                            # (150, 200): two row bands per map (round 6)
    n = d * b * q
    src = torch.randn(n, h, w, generator=g).to(device).requires_grad_()
    idx = torch.randperm(n, generator=g).to(torch.int32).to(device)          # every map once, any order
    coords = torch.rand(d * b, p, 2, generator=g).to(device)
    cidx = (idx // q).to(torch.int32)
    gout = torch.randn(n, p, generator=g).to(device)
    (ops.point_sample(src, idx, coords, cidx) * gout).sum().backward()
    want = src.grad.view(d, b, q, h * w).permute(1, 0, 2, 3).to(dt)
    src2 = src.detach().clone().requires_grad_()
    sink = ops.StackGradSink(d, b, q, dt, device)
    (ops.point_sample(src2, idx, coords, cidx, grad_sink=sink) * gout).sum().backward()
    assert sink.grad is not None and sink.grad.dtype == dt and tuple(sink.grad.shape) == (b, d, q, h * w)
    assert torch.equal(sink.grad, want)
    assert float(src2.grad.abs().max()) == 0.0            # the token: zeros of the stack's shape
    # a subset of the maps: the sink does not apply, the ordinary gradient is returned
    src3 = src.detach().clone().requires_grad_()
    sink3 = ops.StackGradSink(d, b, q, dt, device)
    (ops.point_sample(src3, idx[:7], coords, cidx[:7], grad_sink=sink3) * gout[:7]).sum().backward()
    assert sink3.grad is None and float(src3.grad.abs().max()) > 0.0
