"""Shared tiny configurations for the tests (same keyword names as MaskBevModule / the YAML files)."""
import torch


def tiny_kwargs(nx=80, ny=80, vs=0.25, q=8, p=8, c=32, e=48, f=128, ws=5, pc_dim=4):
    return dict(x_range=(-nx * vs / 2, nx * vs / 2), y_range=(-ny * vs / 2, ny * vs / 2), z_range=(-3, 1),
                voxel_size=vs, num_queries=q, max_num_points=p, encoder_feat_channels=[c, c, c],
                backbone_embed_dim=e, head_feat_channels=f, head_out_channels=f, backbone_window_size=ws,
                pc_point_dim=pc_dim, optimiser_type='adam_w', lr=1e-4, weight_decay=1e-4,
                lr_schedulers_type='plateau', differential_lr=False, differential_lr_scaling=1.0, seed=420)


def random_scans(kw, sizes, seed, spread=1.15):
    g = torch.Generator().manual_seed(seed)
    dim = kw.get('pc_point_dim', 4)
    lo = torch.tensor([kw['x_range'][0], kw['y_range'][0], kw['z_range'][0]], dtype=torch.float32)
    hi = torch.tensor([kw['x_range'][1], kw['y_range'][1], kw['z_range'][1]], dtype=torch.float32)
    out = []
    for n in sizes:
        pts = torch.rand(n, dim, generator=g)
        pts[:, :3] = (lo + hi) / 2 + (pts[:, :3] * 2 - 1) * (hi - lo) / 2 * spread
        out.append(pts)
    return out


def random_gt(kw, batch, n_inst, seed):
    """GT in the dataset's format: padded to num_queries, label 1 = object, 0 = padding."""
    g = torch.Generator().manual_seed(seed)
    q = kw['num_queries']
    nx = int((kw['x_range'][1] - kw['x_range'][0]) / kw['voxel_size'])
    ny = int((kw['y_range'][1] - kw['y_range'][0]) / kw['voxel_size'])
    labels = torch.zeros(batch, q, dtype=torch.long)
    masks = torch.zeros(batch, q, ny, nx)
    for b in range(batch):
        for i in range(n_inst):
            h, w = int(torch.randint(4, ny // 4, (1,), generator=g)), int(torch.randint(4, nx // 4, (1,), generator=g))
            y0, x0 = int(torch.randint(0, ny - h, (1,), generator=g)), int(torch.randint(0, nx - w, (1,), generator=g))
            masks[b, i, y0:y0 + h, x0:x0 + w] = 1
            labels[b, i] = 1
    return labels, masks
