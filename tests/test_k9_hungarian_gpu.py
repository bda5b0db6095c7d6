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


@pytest.mark.parametrize('q,g_slots', [(100, 100), (100, 128), (40, 40), (7, 9), (200, 200), (300, 300), (150, 260), (129, 129)])
def test_hungarian_padded_columns_equal_the_square_solve(device, q, g_slots):
    """ops.hungarian(real_cols=K): the dataset's zero-padded instance list makes columns K .. G-1 identical, and K9
    solves the rectangular problem of the K real columns instead of the square one (mbv_hungarian_padded).  Against
    scipy on the full matrix: a permutation into G slots, the same optimal cost (f64 sums, rel 1e-9), and — generic
    real costs, unique optimum — exactly scipy's pairs on the real columns; predictions left over take the padded
    columns in ascending order.  Problems with K = 0, K = G (no padding: the plain solve) and K of every size between.
    200 / 300 queries (BASELINE configs[3] / [4]): the wide kernel's padded mode, the real columns' block in LDS — and K
    beyond what 128 KB of it hold (K = 299 of 300: the plain wide solve from global memory)."""
    from mask_bev_amd import ops
    gen = torch.Generator().manual_seed(q * 7 + g_slots)
    ks = [0, 1, min(q, g_slots) // 3, min(q, g_slots) - 1, min(q, g_slots), 5, 17 % (min(q, g_slots) + 1)]
    if g_slots == q:
        ks.append(g_slots)                       # no padded column at all
    n = len(ks)
    cost = torch.empty(n, q, g_slots)
    for i, k in enumerate(ks):
        real = torch.randn(q, k, generator=gen) * 3 + torch.rand(q, k, generator=gen)
        pad = (torch.randn(q, 1, generator=gen) * 2 + 1).expand(q, g_slots - k)        # one opt-out cost per prediction
        cost[i] = torch.cat([real, pad], 1)
    got = ops.hungarian(cost.to(device), real_cols=torch.tensor(ks, dtype=torch.int32, device=device)).cpu().numpy()
    for i, k in enumerate(ks):
        c = cost[i].double().numpy()
        rows, cols = linear_sum_assignment(c)
        assert len(set(got[i].tolist())) == q and got[i].min() >= 0 and got[i].max() < g_slots
        ours = float(c[np.arange(q), got[i]].sum())
        assert ours == pytest.approx(float(c[rows, cols].sum()), rel=1e-9, abs=1e-9)
        real_ours = {int(p): int(cc) for p, cc in enumerate(got[i]) if cc < k}
        real_ref = {int(p): int(cc) for p, cc in zip(rows, cols) if cc < k}
        assert real_ours == real_ref
        rest = [int(cc) for cc in got[i] if cc >= k]
        if k < g_slots and q == g_slots:          # (non-square problems take the plain solve: any padded slots)
            assert rest == list(range(k, k + len(rest)))                              # ascending padded slots
