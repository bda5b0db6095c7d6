#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own files.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite read the committed
``.npz`` files, never the reference.  The reference's in-repo arithmetic —

    mask_bev/models/networks/swin/swin.py                       (WindowMSA, ShiftWindowMSA, SwinBlock,
                                                                 SwinBlockSequence, CustomSwinTransformer)
    mask_bev/models/networks/mask2former_head/mask2former_head.py  (_forward_head, forward's decoder loop,
                                                                 loss / _loss_by_feat_single / _get_targets_single)

— is imported UNMODIFIED from /root/reference.  Those files import symbols of mmcv / mmdet / mmengine, which
are not installed and not vendored; this script registers stand-in modules for exactly those symbols
(``_install_shims``), backed by ``torch.nn`` layers with the upstream parameter names and by the oracle's
restatement of the upstream algorithms.  What the vectors pin is therefore the reference's own ~1300 lines;
the stand-ins remain "parity unpinned" (SURVEY.md §8c).

    python tests/golden/make_golden.py
"""
import copy
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import maskbev_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------------------------
# stand-ins for the mm* symbols the two reference files import
# ----------------------------------------------------------------------------------------------
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v


class FFN(nn.Module):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=None, ffn_drop=0.,
                 dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__()
        self.act = 'gelu' if (act_cfg or {}).get('type', 'ReLU') == 'GELU' else 'relu'
        a = nn.GELU() if self.act == 'gelu' else nn.ReLU()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), a, nn.Dropout(0.)),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.))

    def forward(self, x, identity=None):
        return O.ffn({'f.' + k: v for k, v in self.state_dict().items()}, 'f', x, identity, self.act)


class PatchEmbed(nn.Module):
    def __init__(self, in_channels, embed_dims, conv_type, kernel_size, stride, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.k = kernel_size
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size, stride)
        self.norm = nn.LayerNorm(embed_dims)

    def forward(self, x):
        return O.patch_embed({'p.' + k: v for k, v in self.state_dict().items()}, 'p', x, self.k)


class PatchMerging(nn.Module):
    def __init__(self, in_channels, out_channels, stride=2, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.out_channels, self.stride = out_channels, stride
        self.norm = nn.LayerNorm(4 * in_channels)
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=False)

    def forward(self, x, hw):
        return O.patch_merging({'p.' + k: v for k, v in self.state_dict().items()}, 'p', x, hw, self.stride)


class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats, normalize=True, **kw):
        super().__init__()
        self.num_feats = num_feats

    def forward(self, mask):
        b, h, w = mask.shape
        return O.sine_pos_enc(b, h, w, self.num_feats)


class _MHA(nn.Module):
    """mmcv MultiheadAttention(batch_first=True) around the real nn.MultiheadAttention."""

    def __init__(self, embed_dims, num_heads, **kw):
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, 0.0)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        q = query + query_pos if query_pos is not None else query
        k = key + key_pos if key_pos is not None else key
        out = self.attn(query=q.transpose(0, 1), key=k.transpose(0, 1), value=value.transpose(0, 1),
                        attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0].transpose(0, 1)
        return identity + out


class _DecLayer(nn.Module):
    def __init__(self, self_attn_cfg, cross_attn_cfg, ffn_cfg, **kw):
        super().__init__()
        self.self_attn = _MHA(**self_attn_cfg)
        self.cross_attn = _MHA(**cross_attn_cfg)
        self.embed_dims = self_attn_cfg['embed_dims']
        self.ffn = FFN(**ffn_cfg)
        self.norms = nn.ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(3)])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, self_attn_mask=None,
                cross_attn_mask=None, key_padding_mask=None, **kw):
        q = self.cross_attn(query=query, key=key, value=value, query_pos=query_pos, key_pos=key_pos,
                            attn_mask=cross_attn_mask, key_padding_mask=key_padding_mask)
        q = self.norms[0](q)
        q = self.self_attn(query=q, key=q, value=q, query_pos=query_pos, key_pos=query_pos, attn_mask=self_attn_mask)
        q = self.norms[1](q)
        q = self.ffn(q)
        return self.norms[2](q)


class Mask2FormerTransformerDecoder(nn.Module):
    def __init__(self, num_layers, layer_cfg, return_intermediate=True, init_cfg=None, **kw):
        super().__init__()
        self.num_layers = num_layers
        self.layers = nn.ModuleList([_DecLayer(**layer_cfg) for _ in range(num_layers)])
        self.embed_dims = self.layers[0].embed_dims
        self.post_norm = nn.LayerNorm(self.embed_dims)


class PixelDecoderShim(nn.Module):
    """Parameters with mmdet's MSDeformAttnPixelDecoder names; forward = the oracle's restatement."""

    def __init__(self, cfg_ns, sd):
        super().__init__()
        self.cfg_ns = cfg_ns
        self.keys = [k for k in sd if k.startswith(O.HEAD + 'pixel_decoder.')]
        self.params = nn.ParameterDict({k.replace('.', '/'): nn.Parameter(sd[k].clone()) for k in self.keys})

    def init_weights(self):
        pass

    def forward(self, feats):
        sd = {k: self.params[k.replace('.', '/')] for k in self.keys}
        return O.pixel_decoder_forward(self.cfg_ns, sd, feats)


class LinearPixelDecoderShim(nn.Module):
    """A pixel decoder that owes NOTHING to the oracle: fixed random 1 x 1 linear maps of the backbone features —
    mask_features = W_m feats[0], memories = (W_0 feats[3], W_1 feats[2], W_2 feats[1]) (coarse to fine, the order mmdet's
    MSDeformAttnPixelDecoder returns).  With it the head fixture pins the reference's Mask2FormerHead.forward / loss
    independently of ``O.pixel_decoder_forward`` (the first fixture's stand-in IS the oracle's restatement)."""

    def __init__(self, in_channels, feat, out, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.wm = nn.Parameter(torch.randn(out, in_channels[0], generator=g) / math.sqrt(in_channels[0]))
        self.wl = nn.ParameterList([nn.Parameter(torch.randn(feat, c, generator=g) / math.sqrt(c))
                                    for c in (in_channels[3], in_channels[2], in_channels[1])])

    def init_weights(self):
        pass

    def forward(self, feats):
        mask_features = torch.einsum('oc,bchw->bohw', self.wm, feats[0])
        memories = [torch.einsum('oc,bchw->bohw', w, f) for w, f in zip(self.wl, (feats[3], feats[2], feats[1]))]
        return mask_features, memories


class InstanceData:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Assigner:
    def __init__(self, cfg_ns):
        self.cfg_ns = cfg_ns

    def assign(self, pred_instances, gt_instances, img_meta=None):
        from scipy.optimize import linear_sum_assignment
        cost = O.match_cost(self.cfg_ns, pred_instances.scores, pred_instances.masks, gt_instances.labels,
                            gt_instances.masks).detach().cpu()
        r, c = linear_sum_assignment(cost)
        gt_inds = torch.zeros(pred_instances.scores.shape[0], dtype=torch.long)
        gt_inds[torch.from_numpy(r)] = torch.from_numpy(c) + 1
        return types.SimpleNamespace(gt_inds=gt_inds)


class _Sampler:
    def sample(self, assign_result, pred_instances, gt_instances):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return types.SimpleNamespace(pos_inds=pos, neg_inds=neg, pos_assigned_gt_inds=assign_result.gt_inds[pos] - 1,
                                     avg_factor=len(pos) + len(neg))


class _LossCls(nn.Module):
    def __init__(self, class_weight, loss_weight):
        super().__init__()
        self.class_weight, self.loss_weight = class_weight, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None):
        ce = F.cross_entropy(cls_score, label, weight=cls_score.new_tensor(self.class_weight), reduction='none')
        if weight is not None:
            ce = ce * weight.float()
        return self.loss_weight * ce.sum() / (avg_factor + torch.finfo(torch.float32).eps)


class _LossMask(nn.Module):
    def forward(self, pred, target, avg_factor=None):
        bce = F.binary_cross_entropy_with_logits(pred, target.float(), reduction='none')
        return 5.0 * bce.sum() / (avg_factor + torch.finfo(torch.float32).eps)


class _LossDice(nn.Module):
    def forward(self, pred, target, avg_factor=None):
        p = pred.sigmoid().flatten(1)
        t = target.flatten(1).float()
        d = (2 * (p * t).sum(1) + 1.0) / (p.sum(1) + t.sum(1) + 1.0)
        return 5.0 * (1 - d).sum() / (avg_factor + torch.finfo(torch.float32).eps)


def get_uncertain_point_coords_with_randomness(mask_preds, labels, num_points, oversample_ratio,
                                               importance_sample_ratio):
    g = mask_preds.shape[0]
    n_samp = int(num_points * oversample_ratio)
    coords = torch.rand(g, n_samp, 2)
    unc = -torch.abs(O.point_sample(mask_preds, coords))
    n_unc = int(importance_sample_ratio * num_points)
    idx = torch.topk(unc[:, 0, :], k=n_unc, dim=1)[1] + (n_samp * torch.arange(g, dtype=torch.long))[:, None]
    coords = coords.view(-1, 2)[idx.view(-1), :].view(g, n_unc, 2)
    if num_points - n_unc > 0:
        coords = torch.cat((coords, torch.rand(g, num_points - n_unc, 2)), dim=1)
    return coords


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def _install_shims(state):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

        def build(self, cfg, default_args=None):
            t = cfg['type'].split('.')[-1]
            if t == 'MSDeformAttnPixelDecoder':
                if state.get('pd_kind') == 'linear':
                    c = state['cfg']
                    e = c.embed_dim
                    state['pd'] = LinearPixelDecoderShim([e, 2 * e, 4 * e, 8 * e], c.head_feat, c.head_out, seed=77)
                    return state['pd']
                return PixelDecoderShim(state['cfg'], state['sd'])
            if t == 'HungarianAssigner':
                return _Assigner(state['cfg'])
            if t == 'MaskPseudoSampler':
                return _Sampler()
            if t == 'CrossEntropyLoss':
                return _LossMask() if cfg.get('use_sigmoid') else _LossCls(cfg['class_weight'], cfg['loss_weight'])
            if t == 'DiceLoss':
                return _LossDice()
            raise KeyError(t)

    reg = _Registry()

    def build_norm_layer(cfg, n):
        assert cfg['type'] == 'LN'
        return 'ln', nn.LayerNorm(n)

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(t, mean, std, a, b)

    mod('mmcv')
    mod('mmcv.cnn', build_norm_layer=build_norm_layer, Conv2d=nn.Conv2d)
    mod('mmcv.cnn.bricks')
    mod('mmcv.cnn.bricks.transformer', FFN=FFN, build_dropout=lambda cfg: nn.Identity())
    mod('mmcv.ops', point_sample=O.point_sample)
    mod('mmdet')

    class AnchorFreeHead(BaseModule):
        pass

    class MaskFormerHead(AnchorFreeHead):
        pass

    mod('mmdet.models', PatchEmbed=PatchEmbed, PatchMerging=PatchMerging, MaskFormerHead=MaskFormerHead,
        AnchorFreeHead=AnchorFreeHead, Mask2FormerTransformerDecoder=Mask2FormerTransformerDecoder,
        SinePositionalEncoding=SinePositionalEncoding)
    mod('mmdet.models.backbones')
    mod('mmdet.models.backbones.swin', swin_converter=lambda x: x)
    mod('mmdet.models.utils', get_uncertain_point_coords_with_randomness=get_uncertain_point_coords_with_randomness,
        multi_apply=multi_apply)
    mod('mmdet.registry', MODELS=reg, TASK_UTILS=reg)
    mod('mmdet.structures', SampleList=list)
    mod('mmdet.utils', reduce_mean=lambda t: t, InstanceList=list)
    mod('mmengine', to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x)
    mod('mmengine.model', BaseModule=BaseModule, ModuleList=nn.ModuleList, caffe2_xavier_init=lambda *a, **k: None)
    mod('mmengine.model.weight_init', trunc_normal_=trunc_normal_,
        trunc_normal_init=lambda m, std=.02, bias=0.: (trunc_normal_(m.weight, std=std), nn.init.constant_(m.bias, bias)
                                                       if m.bias is not None else None),
        constant_init=lambda m, val, bias=0: (nn.init.constant_(m.weight, val), nn.init.constant_(m.bias, bias)))
    mod('mmengine.runner')
    mod('mmengine.runner.checkpoint', _load_checkpoint=None)
    mod('mmengine.structures', InstanceData=InstanceData)


def _load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# ----------------------------------------------------------------------------------------------
def golden_swin(ref_swin, name, hw, in_ch, embed, ws, depths, heads, seed):
    """Reference CustomSwinTransformer.forward on seeded input/weights."""
    torch.manual_seed(seed)
    h, w = hw
    net = ref_swin.CustomSwinTransformer(pretrain_img_size=(w, h), in_channels=in_ch, embed_dims=embed, patch_size=4,
                                         window_size=ws, mlp_ratio=4, depths=depths, num_heads=heads,
                                         strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None,
                                         patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                                         use_abs_pos_embed=True, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'),
                                         with_cp=False, init_cfg=None, swap_dims=False)
    with torch.no_grad():
        for n_, p in net.named_parameters():          # non-trivial values everywhere (biases, LN, tables)
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.2) + (1.0 if 'norm' in n_ and 'weight' in n_ else 0.0))
    x = torch.randn(2, in_ch, h, w)
    with torch.no_grad():
        outs = net(x)
    out = {'x': x, 'cfg_hw': np.array([h, w]), 'cfg': np.array([in_ch, embed, ws] + list(depths) + list(heads))}
    out.update({'sd.' + k: v for k, v in net.state_dict().items()})
    out.update({f'out{i}': o for i, o in enumerate(outs)})
    np.savez_compressed(os.path.join(HERE, name), **_np(out))
    print(name, [tuple(o.shape) for o in outs])


def golden_head(ref_head_mod, name, seed, with_loss=True):
    """Reference Mask2FormerHead.forward (+ loss) on seeded features/weights."""
    from mask_bev.utils.config import Config   # importable part of the reference
    kw = dict(x_range=(-8, 8), y_range=(-8, 8), z_range=(-3, 1), voxel_size=0.25, num_queries=6, max_num_points=4,
              encoder_feat_channels=[8, 8, 8], backbone_embed_dim=8, head_feat_channels=32, head_out_channels=32,
              pd_layers=2, pd_heads=4, pd_ffn=48, dec_layers=4, dec_heads=4, dec_ffn=40, num_points=96)
    cfg = O.make_cfg(**kw)
    sd = O.make_state_dict(cfg, seed)
    STATE['cfg'], STATE['sd'] = cfg, sd
    e = cfg.embed_dim
    in_ch = [e, 2 * e, 4 * e, 8 * e]
    head_cfg = Config(dict(
        in_channels=in_ch, strides=[4, 8, 16, 32], feat_channels=32, out_channels=32, num_things_classes=1,
        num_stuff_classes=0, num_queries=6, num_transformer_feat_level=3,
        pixel_decoder=dict(type='mmdet.MSDeformAttnPixelDecoder', num_outs=3,
                           encoder=dict(num_layers=2, layer_cfg=dict(self_attn_cfg=dict(num_levels=3, num_heads=4)))),
        enforce_decoder_input_project=False, positional_encoding=dict(num_feats=16, normalize=True),
        transformer_decoder=dict(return_intermediate=True, num_layers=4, layer_cfg=dict(
            self_attn_cfg=dict(embed_dims=32, num_heads=4), cross_attn_cfg=dict(embed_dims=32, num_heads=4),
            ffn_cfg=dict(embed_dims=32, feedforward_channels=40, act_cfg=dict(type='ReLU')))),
        loss_cls=dict(type='mmdet.CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0, class_weight=cfg.class_weight),
        loss_mask=dict(type='mmdet.CrossEntropyLoss', use_sigmoid=True, loss_weight=5.0),
        loss_dice=dict(type='mmdet.DiceLoss', loss_weight=5.0),
        train_cfg=dict(num_points=96, oversample_ratio=3.0, importance_sample_ratio=0.75,
                       assigner=dict(type='mmdet.HungarianAssigner'), sampler=dict(type='mmdet.MaskPseudoSampler'))))

    class _Cfg(AttrDict):
        pass

    head = ref_head_mod.Mask2FormerHead(**{k: (_Cfg(v) if isinstance(v, dict) else v) for k, v in head_cfg.items()})
    # load the oracle-layout weights into the reference module (pixel decoder shim already holds its own)
    own = head.state_dict()
    loaded = 0
    for k in own:
        src = O.HEAD + k
        if src in sd:
            own[k] = sd[src].clone()
            loaded += 1
    head.load_state_dict(own)
    assert loaded == len([k for k in own if not k.startswith('pixel_decoder.')]), (loaded, len(own))
    torch.manual_seed(seed + 1)
    feats = [torch.randn(2, in_ch[i], 64 // 2 ** (i + 2), 64 // 2 ** (i + 2)) for i in range(4)]
    metas = [Config(dict(metainfo={})) for _ in range(2)]
    with torch.no_grad():
        cls_list, mask_list, _ = head.forward(feats, metas)
    out = {f'feat{i}': f for i, f in enumerate(feats)}
    out.update({'sd.' + k: v for k, v in sd.items() if k.startswith(O.HEAD)})
    if STATE.get('pd_kind') == 'linear':           # the stand-in's own maps travel with the fixture
        out['pd_lin.wm'] = STATE['pd'].wm
        for i, w in enumerate(STATE['pd'].wl):
            out[f'pd_lin.w{i}'] = w
    out.update({f'cls{i}': c for i, c in enumerate(cls_list)})
    out.update({f'mask{i}': m for i, m in enumerate(mask_list)})
    if with_loss:
        labels = torch.zeros(2, 6, dtype=torch.long)
        labels[:, :2] = 1
        masks = torch.zeros(2, 6, 64, 64)
        masks[0, 0, 5:20, 8:30] = 1
        masks[0, 1, 40:60, 30:50] = 1
        masks[1, 0, 10:30, 10:20] = 1
        masks[1, 1, 35:45, 5:60] = 1
        torch.manual_seed(seed + 2)                 # the reference draws its points from the global RNG
        ld = head.loss(cls_list, mask_list, labels, masks, [{} for _ in range(2)], [None] * len(cls_list), [None, None])
        out['labels_gt'], out['masks_gt'] = labels, masks
        out['loss_keys'] = np.array(list(ld.keys()))
        out['loss_vals'] = np.array([float(v) for v in ld.values()], dtype=np.float64)
        out['loss_seed'] = np.array(seed + 2)
    np.savez_compressed(os.path.join(HERE, name), **_np(out))
    print(name, len(cls_list), tuple(mask_list[0].shape), 'loss terms', len(ld) if with_loss else 0)


STATE = {}

if __name__ == '__main__':
    assert os.path.isdir(REF), 'the reference tree is only present in the build container'
    _install_shims(STATE)
    sys.path.insert(0, REF)
    swin = _load_ref('ref_swin', 'mask_bev/models/networks/swin/swin.py')
    golden_swin(swin, 'swin_square_40.npz', (40, 40), 6, 8, 5, (2, 2, 2, 2), (1, 2, 2, 4), seed=1)
    golden_swin(swin, 'swin_nonsquare_pad_44x36.npz', (44, 36), 5, 8, 4, (2, 2, 2, 2), (1, 2, 4, 8), seed=2)
    head_mod = _load_ref('ref_m2f_head', 'mask_bev/models/networks/mask2former_head/mask2former_head.py')
    golden_head(head_mod, 'mask2former_head_q6.npz', seed=3)
    STATE['pd_kind'] = 'linear'
    golden_head(head_mod, 'mask2former_head_linpd_q6.npz', seed=5)
