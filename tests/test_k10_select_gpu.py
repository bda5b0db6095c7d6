"""K10 radix-select importance sampling vs torch.topk: identical SET of selected points (incl. ties)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('R,n,k', [(7, 37632, 9408), (3, 600, 150), (2, 513, 513), (4, 1000, 1), (5, 2048, 1024)])
def test_select_matches_topk(device, R, n, k):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(n + k)
    logits = torch.randn(R, n, generator=g) * 4
    logits[0, : n // 3] = logits[0, : n // 3].round()            # many exact ties, also around the threshold
    logits[-1] = 0.25                                            # a constant row: every key ties
    coords = torch.rand(R, n, 2, generator=g)
    out = ops.select_uncertain_points(logits.to(device), coords.to(device), k).cpu()
    idx = torch.topk(-logits.abs(), k=k, dim=1)[1]
    for r in range(R):
        thr = logits[r].abs().kthvalue(k).values
        strictly = (logits[r].abs() < thr).nonzero().flatten()
        equal = (logits[r].abs() == thr).nonzero().flatten()
        want = torch.cat([strictly, equal[: k - strictly.numel()]]).sort().values       # ties → lowest indices
        assert torch.equal(out[r], coords[r][want])
        # same multiset of |logit| values as torch.topk
        assert torch.equal(logits[r].abs()[want].sort().values, logits[r].abs()[idx[r]].sort().values)


@pytest.mark.parametrize('R,n,k,H,W,n_rand', [(6, 37632, 9408, 128, 128, 3136), (3, 600, 150, 24, 20, 50),
                                              (2, 513, 513, 7, 9, 0), (4, 40960, 1, 128, 128, 5),
                                              (3, 40960, 16384, 64, 64, 3),      # the most the fused kernel places in LDS
                                              (2, 40000, 20000, 64, 64, 7)])     # beyond it: the two-kernel form
def test_fused_sample_select_equals_two_kernel_form(device, R, n, k, H, W, n_rand):
    """The fused importance sampling returns exactly what K8 followed by K10 (+ cat of the uniform tail) returns."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(n + k + H)
    src = (torch.randn(R + 2, H, W, generator=g) * 5).to(device)
    src[1] = src[1].round()                                        # plateaus → exact ties in the sampled logits
    idx = torch.tensor([(3 * i + 1) % (R + 2) for i in range(R)], dtype=torch.int32, device=device)
    coords = torch.rand(R, n, 2, generator=g).to(device)
    rand_c = torch.rand(R, n_rand, 2, generator=g).to(device) if n_rand else None
    rows = torch.arange(R, dtype=torch.int32, device=device)
    logits = ops.point_sample(src, idx, coords, rows)
    want = ops.select_uncertain_points(logits, coords, k)
    if rand_c is not None:
        want = torch.cat((want, rand_c), 1)
    got = ops.sample_select_uncertain(src, idx, coords, k, rand_c)
    assert got.shape == want.shape
    assert torch.equal(got, want)


def test_fused_sample_select_in_kernel_points(device):
    """Seed form: the candidates generated inside the kernel are exactly uniform_points(seed); the selection equals
    the two-kernel form on them; the generator is uniform and changes with the seed and the row."""
    from mask_bev_amd import ops
    R, n, k, H, W, n_rand = 5, 37632, 9408, 128, 128, 3136
    g = torch.Generator().manual_seed(3)
    src = (torch.randn(R, H, W, generator=g) * 5).to(device)
    idx = torch.arange(R, dtype=torch.int32, device=device)
    rand_c = torch.rand(R, n_rand, 2, generator=g).to(device)
    seed = torch.tensor([0x1234_5678_9ABC_DEF], dtype=torch.int64, device=device)
    coords = ops.uniform_points(seed, R, n)
    assert coords.shape == (R, n, 2) and float(coords.min()) >= 0.0 and float(coords.max()) < 1.0
    assert abs(float(coords.mean()) - 0.5) < 2e-3 and abs(float(coords.var()) - 1.0 / 12.0) < 1e-3
    hist = torch.histc(coords[..., 0].flatten(), bins=64, min=0, max=1)
    assert float((hist - hist.mean()).abs().max() / hist.mean()) < 0.10          # 5 sigma of a 2940-count bin
    cx = coords[0, :, 0] - 0.5
    assert abs(float((cx[:-1] * cx[1:]).mean()) * 12.0) < 0.02                 # no lag-1 correlation
    assert not torch.equal(coords[0], coords[1])
    seed2 = seed + 1
    assert not torch.equal(ops.uniform_points(seed2, R, n), coords)
    want = ops.sample_select_uncertain(src, idx, coords, k, rand_c)
    got = ops.sample_select_uncertain(src, idx, None, k, rand_c, seed=seed, num_candidates=n)
    assert torch.equal(got, want)
