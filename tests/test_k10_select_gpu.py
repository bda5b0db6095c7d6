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
                                              (2, 513, 513, 7, 9, 0), (4, 40960, 1, 128, 128, 5)])
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
