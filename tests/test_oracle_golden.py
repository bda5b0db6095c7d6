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


def test_mask2former_head_matches_reference_with_independent_pixel_decoder():
    """The same reference class on a second fixture whose pixel-decoder stand-in owes nothing to the oracle (fixed random
    1 x 1 linear maps of the backbone features, tests/golden/make_golden.py LinearPixelDecoderShim; the first fixture's
    stand-in is ``O.pixel_decoder_forward`` itself): forward of all five decoder outputs and all 20 loss terms."""
    g = _load('mask2former_head_linpd_q6.npz')
    cfg = _head_cfg()
    sd = {k[3:]: v for k, v in g.items() if k.startswith('sd.')}
    feats = [g[f'feat{i}'] for i in range(4)]

    def linear_pixel_decoder(fs):
        mf = torch.einsum('oc,bchw->bohw', g['pd_lin.wm'], fs[0])
        return mf, [torch.einsum('oc,bchw->bohw', g[f'pd_lin.w{i}'], f) for i, f in enumerate((fs[3], fs[2], fs[1]))]

    cls_list, mask_list, _ = O.head_forward(cfg, sd, feats, pixel_decoder=linear_pixel_decoder)
    for i in range(5):
        torch.testing.assert_close(cls_list[i], g[f'cls{i}'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mask_list[i], g[f'mask{i}'], rtol=1e-4, atol=2e-4)
    torch.manual_seed(int(g['loss_seed']))
    ld = O.loss_dict(cfg, [g[f'cls{i}'] for i in range(5)], [g[f'mask{i}'] for i in range(5)], g['labels_gt'], g['masks_gt'],
                     O.PointSource(None))
    keys = [str(k) for k in g['loss_keys']]
    assert list(ld.keys()) == keys
    for k, ref in zip(keys, g['loss_vals'].tolist()):
        assert float(ld[k]) == pytest.approx(ref, rel=1e-5, abs=1e-6), k


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


def test_batch_oracle_matches_reference_transforms():
    """oracle/batch_oracle.py vs the reference's own FilterSmallMasks + MaskToLabelInstanceMasks
    (tests/golden/instance_masks.npz, generated by tests/golden/make_golden_batch.py)."""
    import numpy as np
    from oracle import batch_oracle as BO
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'instance_masks.npz'))
    for name in ('a', 'b', 'c', 'd'):
        q, minpix = (int(v) for v in z[f'{name}_cfg'])
        labels, masks, inst = BO.instance_targets(z[f'{name}_map'], q, minpix)
        ref_labels, ref_masks, ref_order = z[f'{name}_labels'], z[f'{name}_masks'], z[f'{name}_order'].tolist()
        assert np.array_equal(labels, ref_labels)                      # CAR for the first n slots, 0 after
        assert sorted(ref_order) == inst                               # same instances survive the pixel filter
        for slot, i in enumerate(inst):                                # same mask per instance, whatever the slot
            assert np.array_equal(masks[slot].astype(np.uint8), ref_masks[ref_order.index(i)])
        assert not masks[len(inst):].any() and not ref_masks[len(inst):].any()


def test_mask_iou_oracle_matches_reference_function():
    """oracle/metrics_oracle.batched_mask_iou vs the reference's own (tests/golden/mask_iou.npz)."""
    import numpy as np
    from oracle import metrics_oracle as MO
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mask_iou.npz'))
    got = MO.batched_mask_iou(torch.from_numpy(z['masks1']).float(), torch.from_numpy(z['masks2']).bool())
    assert torch.allclose(got, torch.from_numpy(z['iou']), rtol=0, atol=0)


def test_fourier_encoding_matches_reference_class():
    """A3: O.fourier_encode against the reference's own LearnableFourierPositionalEncoding (tests/golden/fourier.npz,
    written by tests/golden/make_golden_fourier.py from the unmodified class), both group settings."""
    import numpy as np
    z = np.load(os.path.join(GOLD, 'fourier.npz'))
    for g in (1, 2):
        sd = {O.ENC + '_pos_encoder.' + k: torch.from_numpy(z[f'g{g}_{k}'])
              for k in ('Wr.weight', 'mlp.0.weight', 'mlp.0.bias', 'mlp.2.weight', 'mlp.2.bias')}
        y = O.fourier_encode(sd, torch.from_numpy(z[f'g{g}_x']))
        assert y.shape == (37, 128)
        assert torch.allclose(y, torch.from_numpy(z[f'g{g}_y']), rtol=1e-5, atol=1e-6)
