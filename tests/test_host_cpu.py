"""CPU tests: the C-ABI library loads and exports every symbol include/maskbev_hip.h declares, host-side logic
of the boundary (constructor keys, state_dict layout, registry names, error behaviour), oracle known answers."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import maskbev_oracle as O
from tests.util_cfg import tiny_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from mask_bev_amd import _lib, build
    build.build()
    header = open(os.path.join(ROOT, 'include', 'maskbev_hip.h')).read()
    declared = set(re.findall(r'\b(mbv_[a-z0-9_]+)\s*\(', header))
    assert declared and declared == set(_lib.SIGNATURES.keys())
    lib = _lib.load()                      # binds all of them, raises on a missing symbol
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mbv_abi_version() == _lib.ABI_VERSION
    assert lib.mbv_voxelize_workspace_bytes(120000, 4, 512 * 512) > 4 * 4 * 120000


def test_product_library_reads_no_environment():
    """The shipped library selects no code path from the process environment: it imports no
    `getenv` / `secure_getenv`, holds no `MBV_*` variable name, and the package's path selectors live in
    mask_bev_amd/switches.py, which only `load_from_env()` (A/B scripts) connects to `os.environ`."""
    import subprocess
    from mask_bev_amd import build, switches
    lib = build.build()
    syms = subprocess.run(['nm', '-D', lib], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in syms
    blob = open(lib, 'rb').read()
    assert not re.findall(rb'MBV_[A-Z0-9_]{3,}', blob)
    pkg = os.path.join(ROOT, 'mask_bev_amd')
    for name in sorted(os.listdir(pkg)):
        if name.endswith('.py') and name not in ('switches.py', 'build.py'):
            src = open(os.path.join(pkg, name)).read()
            assert not re.findall(r"environ[^\n]*MBV_", src), name
    assert switches.get('decoder_fused') == 'auto' and switches.defaults() == switches._values
    with switches.override(decoder_fused='0', gemm16='all', shared_kv='0'):
        assert switches.get('decoder_fused') == '0' and switches.get('gemm16') == 'all' and switches.get('shared_kv') is False
    assert switches.get('decoder_fused') == 'auto' and switches.get('gemm16') == 'auto' and switches.get('shared_kv') is True
    from mask_bev_amd import decoder_fused as DF
    assert DF.enabled(torch.bfloat16) and DF.enabled(torch.float16) and not DF.enabled(torch.float32)
    with switches.override(decoder_fused='1'):
        assert DF.enabled(torch.float32)
    with pytest.raises(KeyError):
        switches.set_value('no_such_switch', 1)


def test_product_path_refuses_cpu_tensors():
    from mask_bev_amd import ops
    from mask_bev_amd._lib import MaskBevHipError
    geom = ops.VoxelGeometry.from_ranges([-1, -1, -1, 1, 1, 1], [0.5, 0.5, 2])
    assert geom.grid == [4, 4, 1]
    with pytest.raises(MaskBevHipError):
        ops.voxelize([torch.zeros(10, 4)], geom, 4, 100)
    with pytest.raises(MaskBevHipError):
        ops.window_attention(torch.zeros(1, 5, 5, 12), torch.zeros(12), torch.zeros(81, 1), 1, 5, 0)


def test_module_constructor_state_dict_and_config_contract(tmp_path):
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = tiny_kwargs()
    extra = dict(checkpoint=None, num_workers=4, pin_memory=True, remove_unseen=True, shuffle_train=True,
                 min_num_points=1, augmentations={}, dataset='semantic-kitti', limit_train_batches=1.0,
                 log_every_n_steps=50, min_num_inst_pixels=10)             # YAML keys the module must swallow
    m = MaskBevModule.from_config({**kw, **extra})
    sd_ref = O.make_state_dict(O.make_cfg(**kw))
    sd = m.state_dict()
    assert set(sd) == set(sd_ref)
    assert all(tuple(sd[k].shape) == tuple(sd_ref[k].shape) for k in sd)
    # sub-module attribute names = checkpoint keys
    assert hasattr(m._encoder, '_voxel_layer') and hasattr(m._encoder, '_voxel_encoder')
    assert hasattr(m._encoder, '_middle_encoder') and hasattr(m._encoder, '_layer_norm')
    assert hasattr(m._backbone, '_backbone') and hasattr(m._panoptic_head, '_panoptic_head')
    opt = m.configure_optimizers()
    assert set(opt) == {'optimizer', 'lr_scheduler', 'monitor', 'interval'} and opt['monitor'] == 'train_loss'
    assert isinstance(opt['optimizer'], torch.optim.AdamW)
    with pytest.raises(RuntimeError, match='Invalid batch'):
        m.training_step((1, 2, 3, 4), 0)
    with pytest.raises(ValueError, match='Could not load checkpoint'):
        MaskBevModule.from_config({**kw, 'checkpoint': str(tmp_path / 'nope.ckpt')})
    # checkpoint round trip through the Lightning-style file layout
    path = tmp_path / 'last.ckpt'
    torch.save({'state_dict': sd, 'hyper_parameters': {}}, path)
    m2 = MaskBevModule.from_config({**kw, 'checkpoint': 'last'}, tmp_path)
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), sd.values()))
    assert m.loss({'loss_a': torch.tensor(1.0), 'd0.loss_b': torch.tensor(2.0), 'other': torch.tensor(5.0)}) == 3.0


def test_registry_resolves_reference_type_strings():
    from mask_bev_amd.registry import MODELS, TASK_UTILS
    for name in ['Voxelization', 'PillarFeatureNet', 'PointPillarsScatter', 'CustomSwinTransformer', 'Mask2FormerHead',
                 'mmdet.MSDeformAttnPixelDecoder', 'mmdet.CrossEntropyLoss', 'mmdet.DiceLoss']:
        assert MODELS.get(name) is not None
    for name in ['mmdet.HungarianAssigner', 'mmdet.ClassificationCost', 'mmdet.CrossEntropyLossCost', 'mmdet.DiceCost',
                 'mmdet.MaskPseudoSampler']:
        assert TASK_UTILS.get(name) is not None
    with pytest.raises(KeyError):
        MODELS.get('mmdet.DoesNotExist')


def test_relative_position_index_closed_form():
    from mask_bev_amd.swin import relative_position_index
    for ws in (4, 5, 7, 10):
        assert torch.equal(relative_position_index(ws), O.rel_position_index(ws))


def test_grid_sizes_of_all_workloads():
    """int((max-min)/vs) (mask_bev_module.py:68-69) and mmcv's round() agree for every named workload."""
    from mask_bev_amd import synthetic
    want = {'semantic_kitti_512': (512, 512), 'kitti_496x432': (432, 496), 'waymo_1024': (1024, 1024)}
    for name, (nx, ny) in want.items():
        cfg = O.make_cfg(**synthetic.module_kwargs(name, 1))
        assert (cfg.nx, cfg.ny) == (nx, ny) and cfg.grid3 == [nx, ny, 1]


# ---------------------------------------------------------------- oracle known answers (A1/A2)
def _vox(points, P=2, max_voxels=100):
    cfg = O.make_cfg(x_range=(0, 4), y_range=(0, 4), z_range=(-1, 1), voxel_size=1.0, num_queries=1, max_num_points=P,
                     encoder_feat_channels=[4], backbone_embed_dim=4, head_feat_channels=4, head_out_channels=4,
                     max_voxels=max_voxels)
    pts = torch.tensor(points, dtype=torch.float32)
    return O.voxelize(cfg, [pts]), O.voxelize(cfg, [pts], use_c=False)


def test_voxelize_known_answers():
    pts = [[0.5, 0.5, 0, 1], [3.5, 0.5, 0, 2], [0.6, 0.4, 0, 3], [0.7, 0.3, 0, 4],     # third point of cell (0,0) dropped
           [0.0, 0.5, 0, 5],                      # x == x_min: rejected by the strict pre-filter
           [1.0, 2.0, 0, 6],                      # on a cell border: belongs to the upper cell
           [3.999, 3.999, 0.999, 7], [2.0, 2.0, 1.0, 8]]                                # z == z_max rejected
    (v, n, c), (v2, n2, c2) = _vox(pts)
    assert torch.equal(c, torch.tensor([[0, 0, 0, 0], [0, 0, 0, 3], [0, 0, 2, 1], [0, 0, 3, 3]], dtype=torch.int32))
    assert n.tolist() == [2, 1, 1, 1]
    assert v[0, :, 3].tolist() == [1.0, 3.0] and v[1, :, 3].tolist() == [2.0, 0.0]
    assert torch.equal(v, v2) and torch.equal(n, n2) and torch.equal(c, c2)


def test_voxelize_first_appearance_order_and_cap():
    pts = [[2.5, 2.5, 0, 0], [0.5, 0.5, 0, 1], [2.6, 2.6, 0, 2], [1.5, 0.5, 0, 3]]
    (v, n, c), _ = _vox(pts)
    assert c[:, 2:].tolist() == [[2, 2], [0, 0], [0, 1]]          # order of first appearance, not sorted
    (v, n, c), (v2, n2, c2) = _vox(pts, max_voxels=2)
    assert c[:, 2:].tolist() == [[2, 2], [0, 0]] and n.tolist() == [2, 1]
    assert torch.equal(c, c2)


def test_pfn_legacy_decoration_known_answer():
    """Channels 0-2 become centre offsets (legacy aliasing) and the distance is their norm."""
    cfg = O.make_cfg(x_range=(0, 4), y_range=(0, 4), z_range=(-1, 1), voxel_size=1.0, num_queries=1, max_num_points=2,
                     encoder_feat_channels=[4], backbone_embed_dim=4, head_feat_channels=4, head_out_channels=4)
    voxels = torch.tensor([[[2.25, 1.75, 0.5, 9.0], [2.75, 1.25, -0.5, 7.0]]])
    d = O.pfn_decorate(cfg, voxels, torch.tensor([2]), torch.tensor([[0, 0, 1, 2]]))
    fc0 = torch.tensor([2.25 - 2.5, 1.75 - 1.5, 0.5 - 0.0])
    assert torch.allclose(d[0, 0, :3], fc0) and torch.allclose(d[0, 0, 7:10], fc0)
    assert d[0, 0, 3] == 9.0
    assert torch.allclose(d[0, 0, 4:7], torch.tensor([-0.25, 0.25, 0.5]))           # offset from the pillar mean
    assert torch.allclose(d[0, 0, 10], fc0.norm())


def test_window_partition_roundtrip_and_all_masked_rule():
    x = torch.randn(2, 10, 15, 3)
    assert torch.equal(O._window_reverse(O._window_partition(x, 5), 10, 15, 5), x)
    cfg = O.make_cfg(**tiny_kwargs())
    sd = O.make_state_dict(cfg)
    # a mask feature of all-negative logits blocks every key → the reference un-blocks such rows
    q = torch.zeros(1, cfg.num_queries, cfg.head_feat)
    mf = -torch.ones(1, cfg.head_out, 8, 8)
    _, _, blocked = O.forward_head(cfg, sd, q, mf, (4, 4))
    blocked[torch.where(blocked.sum(-1) == blocked.shape[-1])] = False
    assert not blocked.all(-1).any()


def test_tuned_gemm_table_is_lookup_only_and_cpu_safe():
    """The hipBLASLt solution table ships with validator lines and is never applied without a GPU."""
    from mask_bev_amd import tuning
    assert os.path.isfile(tuning.DEFAULT_TABLE)
    lines = open(tuning.DEFAULT_TABLE).read().splitlines()
    validators = [l for l in lines if l.startswith('Validator,')]
    assert {v.split(',')[1] for v in validators} >= {'PT_VERSION', 'HIPBLASLT_VERSION', 'GCN_ARCH_NAME'}
    assert any('gfx950' in v for v in validators)
    entries = [l for l in lines if l and not l.startswith('Validator,')]
    assert len(entries) > 50 and all(len(e.split(',')) == 4 for e in entries)
    assert not any('Rocblas' in e for e in entries)           # hipBLASLt solutions or Default only
    if not torch.cuda.is_available():
        assert tuning.use_tuned_gemms() is False


def test_reference_import_path_and_launcher_cpu(tmp_path):
    """/root/reference: train_mask_bev.py:12 — ``from mask_bev.mask_bev_module import MaskBevModule`` — resolves to
    this implementation, and so do the module paths the reference's own tests / figure scripts import."""
    import importlib
    import yaml
    from mask_bev.mask_bev_module import MaskBevModule
    from mask_bev_amd.mask_bev_module import MaskBevModule as Native
    assert MaskBevModule is Native
    for mod, name in [('mask_bev.models.encoders.mask_bev_encoders', 'MaskBevEncoder'),
                      ('mask_bev.models.backbones.mask_bev_backbone', 'MaskBevBackbone'),
                      ('mask_bev.models.head.mask_bev_panoptic_head', 'MaskBevPanopticHead'),
                      ('mask_bev.models.networks.swin.swin', 'CustomSwinTransformer'),
                      ('mask_bev.models.networks.mask2former_head.mask2former_head', 'Mask2FormerHead'),
                      ('mask_bev.models.training_types', 'OptimizerType')]:
        assert hasattr(importlib.import_module(mod), name)
    # the launcher parses the reference's command line and YAML, then refuses to run without an MI355X
    import train_mask_bev_amd as launcher
    from mask_bev_amd import synthetic
    cfg = tmp_path / 'smoke.yml'
    cfg.write_text(yaml.safe_dump(dict(synthetic.module_kwargs('smoke_96', 2), dataset='synthetic')))
    with pytest.raises(ValueError, match='Could not find config'):
        launcher.main(['--config', str(tmp_path / 'missing.yml'), '--train'])
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match='MI355X'):
            launcher.main(['--config', str(cfg), '--train', '--synthetic'])


def test_fourier_configuration_builds_with_reference_keys():
    """A3: `encoder_encoding_type: fourier` (mask_bev_encoders.py:51-58) builds, with the reference's parameter names
    (`_encoder._pos_encoder.{Wr,mlp.0,mlp.2}`) and a 128 + 7 channel first PFN layer."""
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = dict(tiny_kwargs(), encoder_encoding_type='fourier', encoder_fourier_enc_group=2)
    m = MaskBevModule(**kw)
    sd = m.state_dict()
    ref = O.make_state_dict(O.make_cfg(**kw))
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in sd)
    assert tuple(sd['_encoder._voxel_encoder.pfn_layers.0.linear.weight'].shape) == (16, 135)
    assert tuple(sd['_encoder._pos_encoder.mlp.2.weight'].shape) == (64, 32)
    with pytest.raises(NotImplementedError):
        MaskBevModule(**dict(tiny_kwargs(), encoder_encoding_type='cosine'))


def test_semantic_kitti_readers_match_reference_readers(tmp_path):
    """F2: the `.bin` / `.label` / `poses.txt` / `calib.txt` / `.npy` readers of mask_bev_amd/batch.py on the sample
    files under tests/golden/semantic_kitti_sample, against what the reference's own reader methods returned for the
    same files (tests/golden/readers.npz, written by tests/golden/make_golden_readers.py)."""
    from mask_bev_amd import batch as B
    d = os.path.join(ROOT, 'tests', 'golden', 'semantic_kitti_sample')
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'readers.npz'))
    assert np.array_equal(B.read_velodyne_bin(os.path.join(d, '000000.bin')), z['scan'])
    poses = B.read_poses(os.path.join(d, 'poses.txt'))
    assert poses.shape == (5, 4, 4) and np.array_equal(poses, z['poses'])
    calib = B.read_calib(os.path.join(d, 'calib.txt'))
    assert set(calib) == {'p0', 'p1', 'p2', 'p3', 'velo_to_cam'}
    for k, v in calib.items():
        assert np.array_equal(v, z['calib_' + k]), k
    sem, inst = B.read_semantic_kitti_label(os.path.join(d, '000000.label'))
    sem, inst = B.apply_learning_map(sem, inst, z['learning_map_lut'])
    assert np.array_equal(sem, z['sem']) and np.array_equal(inst, z['inst'])
    m = (np.arange(12, dtype=np.int32).reshape(3, 4) % 5)
    np.save(tmp_path / 'mask.npy', m)
    assert np.array_equal(B.read_mask_cache(tmp_path / 'mask.npy'), m)
    with pytest.raises(ValueError):
        (tmp_path / 'bad.bin').write_bytes(b'\x00' * 10)
        B.read_velodyne_bin(tmp_path / 'bad.bin')


def test_mask_map_matches_coco_protocol_oracle():
    """F3: MaskMeanAveragePrecision (vectorised COCOeval) against the plain-loop restatement of the published
    algorithm in oracle/metrics_oracle.py, on random multi-image cases with two classes, zero-area padding ground
    truths (the dataset's convention), ties and area ranges; plus the closed-form cases."""
    from mask_bev_amd.metrics import MaskMeanAveragePrecision
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed(0)
    for trial in range(4):
        m = MaskMeanAveragePrecision()
        imgs = []
        for _ in range(3):
            q, ng = 12, 9
            iou = torch.rand(q, ng, generator=g, dtype=torch.float64)
            iou[iou < 0.45] = 0
            iou[:, -2:] = 0                                    # padding ground truths never overlap
            if trial == 1:
                iou = (iou * 4).round() / 4                    # ties
            im = dict(ious=iou, scores=torch.rand(q, generator=g, dtype=torch.float64),
                      pred_labels=torch.randint(0, 2, (q,), generator=g),
                      pred_areas=torch.rand(q, generator=g, dtype=torch.float64) * 12000,
                      gt_labels=torch.cat([torch.ones(ng - 2, dtype=torch.long), torch.zeros(2, dtype=torch.long)]),
                      gt_areas=torch.cat([torch.rand(ng - 2, generator=g, dtype=torch.float64) * 12000,
                                          torch.zeros(2, dtype=torch.float64)]))
            imgs.append(im)
        m.images = imgs
        got = m.compute()
        ref = MO.coco_mask_map([{k: v.numpy() for k, v in im.items()} for im in imgs])
        assert got.keys() == ref.keys()
        for k in ref:
            assert got[k] == pytest.approx(ref[k], abs=1e-12), (trial, k)
    # update() from logits: a prediction that reproduces a ground-truth mask exactly scores IoU 1
    m = MaskMeanAveragePrecision()
    gt = torch.zeros(1, 3, 16, 16)
    gt[0, 0, 2:8, 2:8] = 1
    gt[0, 1, 9:15, 3:12] = 1
    logits = (gt.clone() * 2 - 1) * 10
    m.update(logits, torch.tensor([[0.9, 0.8, 0.1]]), torch.tensor([[1, 1, 0]]), gt, torch.tensor([[1, 1, 0]]))
    assert torch.allclose(torch.diag(m.images[0]['ious'])[:2], torch.ones(2, dtype=torch.float64))
    out = m.compute()
    assert out['map_50'] == pytest.approx(0.5)                 # class 1: AP 1; class 0: the zero-area padding is never matched


def test_workmodel_prices_every_instrumented_entry_point():
    """bench.py prices each C-ABI call from its argument list (mask_bev_amd/workmodel.py: positions follow
    include/maskbev_hip.h).  Every modelled entry point must exist in the binding table, and the models of the entry
    points added in round 2 must return the byte counts of their shapes (known-answer checks)."""
    import ctypes
    from mask_bev_amd import _lib, workmodel as W
    for name in W.MODELS:
        assert name in _lib.SIGNATURES, name
    n = 4 * 256 * 128 * 128
    # GroupNorm forward: bf16 in, bf16 out, f32 (4, 256, 64, 64) map added after up-sampling
    args = (ctypes.c_void_p(1), 1, 4, 256, 128, 128, 32, None, None, 1e-5, ctypes.c_void_p(2), 0, 64, 64, 0,
            ctypes.c_void_p(3), 1)
    k, bound, by, fl = W.MODELS['mbv_groupnorm_fwd'](args)
    assert (k, bound, fl) == ('k_gn_fwd', 'hbm', 0.0) and by == n * 4 + 4 * 256 * 64 * 64 * 4
    # ... without the added map
    args = args[:10] + (ctypes.c_void_p(None),) + args[11:]
    assert W.MODELS['mbv_groupnorm_fwd'](args)[2] == n * 4
    # GroupNorm backward: bf16 dy, bf16 x, bf16 dx
    bwd = (None, 1, None, 1, None, None, None, None, 4, 256, 128, 128, 32, 0, None, 1)
    assert W.MODELS['mbv_groupnorm_bwd'](bwd)[2] == n * 6
    # patch merging: f32 map in, bf16 rows out / bf16 dy, f32 x, f32 dx
    m = 4 * 128 * 128 * 192
    assert W.MODELS['mbv_merge_layernorm_fwd']((None, 4, 128, 128, 192, None, None, 1e-5, None, 1))[2] == m * 6
    assert W.MODELS['mbv_merge_layernorm_bwd']((None, 1, None, None, None, None, 4, 128, 128, 192))[2] == m * 10
    # grouped weight gradients: two products
    L = ctypes.c_int64 * 2
    grp = (None, None, None, L(4096, 400), L(768, 256), L(768, 256), None, None, 2)
    k, bound, by, fl = W.MODELS['mbv_gemm16_tn_group'](grp)
    assert k == 'k_gemm16_tn_group'
    assert fl == 2.0 * (4096 * 768 * 768 + 400 * 256 * 256)
    assert by == (4096 * (768 + 768) + 400 * (256 + 256)) * 2.0 + 2 * (768 * 768 + 256 * 256) * 4.0


def test_workmodel_prices_library_calls_known_answers():
    """The families bench.py prices through its dispatch mode: hipBLASLt GEMMs by operand dtype with
    2 M N K flops, MIOpen convolutions, streaming ATen kernels by the bytes of their tensors; views and allocations
    launch nothing.  The C-ABI models added in round 4 (PFN, K1, K16) against their shapes."""
    import ctypes
    from mask_bev_amd import _lib, workmodel as W
    A = torch.ops.aten
    x, w, b = torch.zeros(100, 64), torch.zeros(32, 64), torch.zeros(32)
    fam, bound, by, fl = W.aten_work(A.addmm.default, (b, x, w.t()), {}, torch.zeros(100, 32), cuda_only=False)
    assert (fam, bound, fl) == ('hipblaslt_f32', 'mfma_f32', 2.0 * 100 * 32 * 64)
    assert by == (100 * 64 + 64 * 32 + 32 + 100 * 32) * 4
    xb = torch.zeros(4, 100, 64, dtype=torch.bfloat16)
    fam, bound, by, fl = W.aten_work(A.bmm.default, (xb, xb.transpose(1, 2)), {}, torch.zeros(4, 100, 100, dtype=torch.bfloat16),
                                     cuda_only=False)
    assert (fam, bound, fl) == ('hipblaslt_16bit', 'mfma', 2.0 * 4 * 100 * 100 * 64)
    img, ker = torch.zeros(2, 8, 16, 16), torch.zeros(4, 8, 3, 3)
    # the out= form names its destination among the arguments: not an operand, counted once
    e, fm, dst = torch.zeros(4, 100, 256), torch.zeros(4, 256, 1024), torch.zeros(4, 100, 1024)
    fam, bound, by, fl = W.aten_work(A.bmm.out, (e, fm), {'out': dst}, dst, cuda_only=False)
    assert (fam, bound) == ('hipblaslt_f32', 'mfma_f32') and fl == 2.0 * 4 * 100 * 1024 * 256
    assert by == 4.0 * (e.numel() + fm.numel() + dst.numel())
    fam, bound, by, fl = W.aten_work(A.convolution.default, (img, ker, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1), {},
                                     torch.zeros(2, 4, 16, 16), cuda_only=False)
    assert (fam, fl) == ('miopen_conv', 2.0 * 2 * 4 * 16 * 16 * 8 * 9)
    fam, bound, by, fl = W.aten_work(A.add.Tensor, (x, x), {}, torch.zeros(100, 64), cuda_only=False)
    assert (fam, bound, by, fl) == ('aten_elementwise', 'hbm', 3 * 100 * 64 * 4, 0.0)
    assert W.aten_work(A.sum.default, (x,), {}, torch.zeros(()), cuda_only=False)[0] == 'aten_reduce'
    assert W.aten_work(A.view.default, (x, [64, 100]), {}, x.view(64, 100), cuda_only=False) is None
    assert W.aten_work(A.empty.memory_format, ([3, 3],), {}, torch.empty(3, 3), cuda_only=False) is None
    # K2b kernels: K rows from the hint, V pillars and U units from the arguments
    _lib.WORK_HINT['pfn_rows'] = 1000
    P = ctypes.c_void_p
    k, bound, by, fl = W.MODELS['mbv_pfn_stats']((P(1), P(2), P(3), P(4), P(5), 200, 64, 32, P(6), None))
    assert (k, bound) == ('k_pfn_stats', 'hbm') and by == (2 * 1000 + 3 * 200) * 64 * 4
    assert W.MODELS['mbv_pfn_stats']((P(1), P(None), P(3), P(4), P(5), 200, 64, 32, P(6), None))[2] == (1000 + 200) * 64 * 4
    assert W.MODELS['mbv_pfn_bwd_bn']((None,) * 12 + (200, 128, 32, None, None))[2] == (3 * 1000 + 4 * 200) * 128 * 4
    assert W.MODELS['mbv_msda_prepare_fwd']((None, None, 1, None, None, 4, 5376, 8, 3, 4, None, None, None))[2] == \
        4 * 5376 * 8 * 3 * 4 * (3 * 2 + 12)


def test_workmodel_add_layernorm_bwd2_known_answer():
    """K12's backward with two gradients of y: bf16 dy2 beside an f32 dy, no residual-path gradient, 16-bit dx copy."""
    import ctypes
    from mask_bev_amd import workmodel as W
    P = ctypes.c_void_p
    args = (P(1), 0, P(2), 1, P(None), 0, P(3), P(4), P(5), P(6), 21504, 256, P(7), P(8), 1)
    k, bound, by, fl = W.MODELS['mbv_add_layernorm_bwd2'](args)
    assert (k, bound, fl) == ('k_add_ln_bwd', 'hbm', 0.0)
    assert by == 21504 * 256 * (4 + 2 + 0 + 4 + 4 + 2)


def test_committed_bench_line_traffic_not_below_algorithmic_bytes():
    """Self-check of the measurement: in the committed bench line of this round every HBM-bound
    family's PMC traffic per launch is at least 0.9 x the algorithmic bytes the work model charges — a model that
    exceeds the measured traffic (round 2: shared ground-truth maps charged once per row) inflates `roofline.frac`.
    The line also carries the contract's objects."""
    import glob
    import json
    rounds = sorted(d for d in glob.glob(os.path.join(ROOT, 'profiles', 'r*')) if glob.glob(os.path.join(d, '*bench_default.json')))
    assert rounds, 'no committed bench line under profiles/'
    files = sorted(glob.glob(os.path.join(rounds[-1], '*bench_default.json')))
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    if 'roofline_all' not in line:
        # round 5 on: the printed line is the summary (< 4 KB, test below); the per-family tables of the same run sit
        # beside it as `*bench_detail.json` (bench.py --detail-out)
        assert len(open(files[-1]).read().strip().splitlines()[-1]) < 4096
        detail = files[-1].replace('bench_default.json', 'bench_detail.json')
        assert os.path.exists(detail), detail
        full = json.load(open(detail))
        assert abs(full['value'] - line['value']) <= 1e-4 * line['value'] and full['roofline']['kernel'] == line['roofline']['kernel']
        line = full
    if os.path.basename(rounds[-1]) >= 'r04':
        # round 4: the table prices the WHOLE step (library GEMMs, MIOpen, ATen kernels, PFN, K1 included): the families'
        # time per step adds up to at least 0.9 of the measured step, and `roofline` is the family that costs the most
        total = sum(r['total_ms_per_step'] for r in line['roofline_all'])
        assert total >= 0.9 * line['ms_per_step'], (total, line['ms_per_step'])
        assert abs(line['roofline_coverage'] - total / line['ms_per_step']) < 1e-6
        assert line['roofline']['kernel'] == max(line['roofline_all'], key=lambda r: r['total_ms_per_step'])['kernel']
        assert {'hipblaslt_16bit', 'aten_elementwise', 'k_pfn_bwd_route', 'k_hungarian'} <= {r['kernel'] for r in line['roofline_all']}
        assert line['cpu_baseline']['sample'].startswith('n = 1 scan') and '3 timed iterations' in line['cpu_baseline']['sample']
        assert line['fp32']['roofline']['kernel'] and line['fp32']['roofline_coverage'] > 0.8
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['roofline']['bound'] in ('hbm', 'mfma') and 0 < line['roofline']['frac'] < 1
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] >= 1
    checked = 0
    for r in line['roofline_all']:
        if r['bound'] != 'hbm' or not r.get('traffic'):
            continue
        if r['kernel'] == 'k_hungarian':
            # K9's padded-column form reads only the REAL ground-truth columns of each cost matrix (their number is a
            # device tensor); the model charges the whole matrix as an upper bound.  One wavefront per problem: the family
            # is latency-bound at 0.000 of HBM peak either way, no fraction is inflated by it.
            assert r['frac'] < 0.01
            continue
        # (library families: what is left of `hipblaslt_f32` in the 16-bit step since K2c took the PFN's Linears are the decoder's
        # few-row products and the PFN's weight gradients — operands of a few MB that the launch in front left in the 256 MB
        # memory-side cache, so a part of their operand bytes never comes from HBM: 0.88 measured)
        floor = 0.8 if r['kernel'].startswith(('hipblaslt_', 'aten_', 'miopen_')) else 0.9
        assert r['traffic'] >= floor * r['algorithmic_bytes'], (r['kernel'], r['traffic'], r['algorithmic_bytes'])
        checked += 1
    assert checked >= 20


def test_bench_line_fits_the_drivers_tail_and_keeps_the_contract():
    """the driver parses the LAST stdout line of bench.py from a bounded tail; round 4's 35.8 KB line
    (`roofline_all` of two dtypes) came back `parsed: null`.  `bench.compact_line` builds the printed line from the full
    record: here from round 4's committed full record (the shape every later run produces), and from one inflated with
    long tables, switches and a collectives schedule — always < 4 096 bytes, always with the contract's keys."""
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, 'profiles', 'r04', 'f_bench_default.json')).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 30_000            # the record that did not parse
    required = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
    line = bench.compact_line(full, 'gpurun_out/bench_detail.json')
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for k in required:
        assert k in line, k
    assert abs(line['value'] - full['value']) < 1e-3 * full['value'] and line['dtype'] == 'bf16'
    assert set(line['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert line['roofline']['kernel'] == full['roofline']['kernel']
    assert abs(line['roofline']['frac'] - line['roofline']['achieved'] / line['roofline']['peak']) < 1e-4
    assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'} and 'rows' not in line['cpu_baseline']
    assert line['fp32']['value'] and line['fp32']['roofline']['kernel'] == 'hipblaslt_f32' and 'roofline_all' not in line['fp32']
    assert 'roofline_all' not in line and line['roofline_detail'] == 'gpurun_out/bench_detail.json'
    assert len(line['roofline_top']) == 5 and line['roofline_top'][0]['kernel'] == line['roofline']['kernel']
    assert 'step_roofline' in line and 0 < line['step_roofline']['frac_hbm'] < 1
    # a pathological record: the optional pieces go, the contract stays
    fat = json.loads(json.dumps(full))
    fat['config']['switches'] = {f'switch_{i}': 'x' * 40 for i in range(60)}
    fat['collectives'] = dict(backend='nccl', bytes_per_step=1, note='n' * 300,
                              schedule=[dict(mark='m' * 30, ms=1.0, mb=2.0) for _ in range(80)])
    fat['cpu_baseline']['rows'] = fat['cpu_baseline']['rows'] * 8
    line = bench.compact_line(fat, None)
    assert len(json.dumps(line)) < 4096
    for k in required:
        assert k in line, k
    assert line['roofline']['frac'] > 0 and line['cpu_baseline']['value'] > 0


def test_level_inputs_node_equals_the_plain_ops():
    """ops.level_inputs (one autograd node for flatten + level-embedding add + positional add + casts,
    mask2former_head.py:518-527) against the four plain ops, values and gradients (CPU: the non-arena branch)."""
    import torch
    from mask_bev_amd import ops
    torch.manual_seed(3)
    mem = torch.randn(2, 8, 3, 5, requires_grad=True)
    lw = torch.randn(3, 8, requires_grad=True)
    pos = torch.randn(1, 15, 8)
    a, k = ops.level_inputs(mem, lw, 1, pos, torch.float32)
    ga, gk = torch.randn_like(a), torch.randn_like(k)
    torch.autograd.backward([a, k], [ga, gk])
    g_mem, g_lw = mem.grad.clone(), lw.grad.clone()
    mem.grad = lw.grad = None
    a2 = mem.flatten(2).transpose(1, 2) + lw[1].view(1, 1, -1)
    k2 = a2 + pos
    torch.autograd.backward([a2, k2], [ga, gk])
    assert torch.equal(a, a2) and torch.equal(k, k2)
    assert torch.allclose(g_mem, mem.grad, atol=1e-6) and torch.allclose(g_lw, lw.grad, atol=1e-5)
    assert float(g_lw[0].abs().max()) == 0.0 and float(g_lw[2].abs().max()) == 0.0
