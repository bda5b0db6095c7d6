"""Pins the oracle against golden vectors produced by the REFERENCE's own swin.py / mask2former_head.py
(tests/golden/make_golden.py, run in the build container).  CPU only; reads no reference file."""
import os

import numpy as np
import pytest
import torch

from oracle import maskbev_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiub' else z[k]) for k in z.files}


@pytest.mark.parametrize('name', ['swin_square_40.npz', 'swin_nonsquare_pad_44x36.npz'])
def test_swin_matches_reference(name):
    """CustomSwinTransformer.forward of the reference (swin.py:745-774, incl. padding, shift masks, the
    (w, h) abs-pos-embed quirk) == oracle.swin_forward, atol 2e-4 on O(1..10) activations."""
    g = _load(name)
    in_ch, embed, ws = [int(v) for v in g['cfg'][:3]]
    depths, heads = tuple(int(v) for v in g['cfg'][3:7]), tuple(int(v) for v in g['cfg'][7:11])
    h, w = [int(v) for v in g['cfg_hw']]
    cfg = O.make_cfg(x_range=(0, w), y_range=(0, h), z_range=(-3, 1), voxel_size=1.0, num_queries=4, max_num_points=4,
                     encoder_feat_channels=[in_ch], backbone_embed_dim=embed, head_feat_channels=32,
                     head_out_channels=32, backbone_window_size=ws, depths=depths, num_heads=heads)
    sd = {O.BB + k[3:]: v for k, v in g.items() if k.startswith('sd.')}
    # the reference's relative_position_index buffer == the oracle's closed form
    for k, v in sd.items():
        if k.endswith('relative_position_index'):
            assert torch.equal(v, O.rel_position_index(ws))
    outs = O.swin_forward(cfg, sd, g['x'])
    assert len(outs) == 4
    for i, o in enumerate(outs):
        ref = g[f'out{i}']
        assert o.shape == ref.shape
        torch.testing.assert_close(o, ref, rtol=1e-4, atol=2e-4)


def _head_cfg():
    return O.make_cfg(x_range=(-8, 8), y_range=(-8, 8), z_range=(-3, 1), voxel_size=0.25, num_queries=6,
                      max_num_points=4, encoder_feat_channels=[8, 8, 8], backbone_embed_dim=8, head_feat_channels=32,
                      head_out_channels=32, pd_layers=2, pd_heads=4, pd_ffn=48, dec_layers=4, dec_heads=4, dec_ffn=40,
                      num_points=96)


def test_mask2former_head_forward_matches_reference():
    """Mask2FormerHead.forward of the reference (mask2former_head.py:474-562: level cycling, _forward_head,
    attention-mask rule) == oracle.head_forward."""
    g = _load('mask2former_head_q6.npz')
    cfg = _head_cfg()
    sd = {k[3:]: v for k, v in g.items() if k.startswith('sd.')}
    feats = [g[f'feat{i}'] for i in range(4)]
    cls_list, mask_list, heights = O.head_forward(cfg, sd, feats)
    assert len(cls_list) == 5 and all(h is None for h in heights)
    for i in range(5):
        torch.testing.assert_close(cls_list[i], g[f'cls{i}'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mask_list[i], g[f'mask{i}'], rtol=1e-4, atol=2e-4)


def test_mask2former_loss_matches_reference():
    """Mask2FormerHead.loss of the reference (mask2former_head.py:246-298,326-426; random points drawn from
    the global RNG in the reference's order) == oracle.loss_dict, every one of the 20 terms."""
    g = _load('mask2former_head_q6.npz')
    cfg = _head_cfg()
    cls_list = [g[f'cls{i}'] for i in range(5)]
    mask_list = [g[f'mask{i}'] for i in range(5)]
    torch.manual_seed(int(g['loss_seed']))
    ld = O.loss_dict(cfg, cls_list, mask_list, g['labels_gt'], g['masks_gt'], O.PointSource(None))
    keys = [str(k) for k in g['loss_keys']]
    assert list(ld.keys()) == keys
    for k, ref in zip(keys, g['loss_vals'].tolist()):
        assert float(ld[k]) == pytest.approx(ref, rel=1e-5, abs=1e-6), k
    assert float(O.total_loss(ld)) == pytest.approx(sum(g['loss_vals'].tolist()), rel=1e-5)
