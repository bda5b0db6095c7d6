"""K9 device Hungarian vs scipy.optimize.linear_sum_assignment: identical assignment on generic (tie-free)
costs, identical optimal COST on matrices with duplicated columns (tied optima)."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,r,c', [(6, 100, 100), (3, 8, 8), (4, 20, 50), (4, 50, 20), (2, 128, 128), (5, 1, 7),
                                   (3, 200, 200), (2, 300, 300), (2, 150, 260), (2, 260, 150), (2, 129, 129),
                                   (1, 320, 320)])
def test_hungarian_matches_scipy(device, n, r, c):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(r * 131 + c)
    cost = torch.randn(n, r, c, generator=g) * 3 + torch.rand(n, r, c, generator=g)
    got = ops.hungarian(cost.to(device)).cpu().numpy()
    for i in range(n):
        rows, cols = linear_sum_assignment(cost[i].numpy())
        want = np.full(r, -1, dtype=np.int64)
        want[rows] = cols
        assert np.array_equal(got[i], want)


def test_hungarian_tied_columns_same_cost(device):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(5)
    base = torch.rand(4, 100, 12, generator=g)
    cost = torch.cat([base, base[:, :, :1].expand(-1, -1, 88)], dim=2).contiguous()   # 88 identical padded columns
    got = ops.hungarian(cost.to(device)).cpu().numpy()
    for i in range(4):
        rows, cols = linear_sum_assignment(cost[i].numpy())
        assert sorted(got[i].tolist()) == list(range(100))                             # a permutation
        ours = float(cost[i].numpy()[np.arange(100), got[i]].sum())
        assert ours == pytest.approx(float(cost[i].numpy()[rows, cols].sum()), rel=1e-6)
        # the queries matched to the 11 distinct real columns agree with scipy
        real_ours = {int(q): int(cc) for q, cc in enumerate(got[i]) if cc in range(1, 12)}
        real_ref = {int(q): int(cc) for q, cc in zip(rows, cols) if cc in range(1, 12)}
        assert real_ours == real_ref
