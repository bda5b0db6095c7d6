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
