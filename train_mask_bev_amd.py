#!/usr/bin/env python3
"""Launcher of the MI355X-native MaskBEV path — the counterpart of /root/reference: train_mask_bev.py:34-119.

Same command line (``--config/-c``, ``--train/-t``, ``--test/-e``), same YAML files (every key of
``configs/training/**.yml`` is accepted; the model keys go to ``MaskBevModule.from_config`` unchanged), same
checkpoint folder convention (``checkpoints/<config stem>/``, ``last.ckpt`` plus the best
``<stem>-epoch=EE-<metric>=V.ckpt``; ``--test`` picks the best file by the ``val_loss=`` / ``train_loss=`` in its
name like :57-64 does).

What differs: the reference hands the loop to ``pytorch_lightning.Trainer(strategy='ddp')`` (:92-112), which is not
installed on the MI355X image.  The loop here is the built-in one: one process per GPU (``torchrun`` / RANK,
WORLD_SIZE), parameters in a :class:`mask_bev_amd.arena.ParameterArena`, the static part of the step replayed from
HIP graphs (:class:`mask_bev_amd.graph.GraphedTrainStep`), gradients averaged over RCCL by
:class:`mask_bev_amd.ddp.GradientAllReducer` underneath the backward, ``ReduceLROnPlateau`` / ``CosineAnnealingLR``
stepped once per epoch on ``train_loss`` as ``configure_optimizers`` declares (mask_bev_module.py:161-166).

Data (the reference's DataModules, /root/reference: train_mask_bev.py:68-83, are outside the hot path):
``dataset: synthetic`` (or ``--synthetic``) draws SemanticKITTI-shaped scans and box masks on the GPU
(mask_bev_amd/synthetic.py); ``dataset: semantic-kitti`` reads ``<root>/sequences/SS/velodyne/*.bin`` with the
instance-map cache ``<root>/sequences/SS/mask_cache/*.npy`` the reference's mask dataset writes
(semantic_kitti_mask_dataset.py:121-137) and builds the (labels, masks) targets on the GPU (batch.py, K14).
"""
from __future__ import annotations

import argparse
import os
import pathlib
import re
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault('OMP_NUM_THREADS', str(6))          # /root/reference: train_mask_bev.py:14

from mask_bev.mask_bev_module import MaskBevModule          # noqa: E402  (the reference's import line, :12)


def get_metric_from_name(path, regex):
    return float(regex.search(str(path)).group(1))


class SyntheticBatches:
    """``len`` batches per epoch of synthetic scans in the reference's batch contract, already on the device."""

    def __init__(self, config, device, rank, batches_per_epoch):
        from mask_bev_amd import synthetic
        self.syn, self.device, self.rank, self.n = synthetic, device, rank, batches_per_epoch
        vs = config['voxel_size']
        self.nx = int((config['x_range'][1] - config['x_range'][0]) / vs)
        self.ny = int((config['y_range'][1] - config['y_range'][0]) / vs)
        self.cfg = config

    def __len__(self):
        return self.n

    def batch(self, epoch, i):
        c, syn = self.cfg, self.syn
        gen = torch.Generator(device=self.device).manual_seed(int(c.get('seed', 420)) + 1000 * self.rank + 100003 * epoch + i)
        b = int(c.get('batch_size', 1))
        scans = []
        for _ in range(b):
            s = syn.lidar_scan(int(c.get('synthetic_points', 120000)), int(c.get('pc_point_dim', 4)), gen, self.device)
            if c['x_range'][0] >= 0:
                s[:, 0] = s[:, 0].abs()
            scans.append(s)
        labels, masks = syn.gt_masks(b, int(c['num_queries']), self.ny, self.nx, gen, self.device, cell=c['voxel_size'])
        return scans, (labels, masks)


class SemanticKittiCacheBatches:
    """``.bin`` scans + the reference's ``.npy`` instance-map cache; targets are expanded on the GPU (K14)."""

    def __init__(self, config, device, rank, world, root, sequences):
        from mask_bev_amd import batch as B
        self.B, self.device = B, device
        files = []
        for seq in sequences:
            d = pathlib.Path(root) / 'sequences' / f'{int(seq):02d}'
            for f in sorted((d / 'velodyne').glob('*.bin')):
                m = d / 'mask_cache' / (f.stem + '.npy')
                if m.exists():
                    files.append((f, m))
        if not files:
            raise ValueError(f'no (velodyne/*.bin, mask_cache/*.npy) pairs under {root}')
        bsz = int(config.get('batch_size', 1))
        usable = len(files) - len(files) % (world * bsz)           # drop_last, equal work per rank
        self.files = files[rank:usable:world]
        self.bsz = bsz
        self.collate = B.InstanceMapCollate(int(config['num_queries']), device,
                                            int(config.get('min_num_inst_pixels', 0)))
        self.shuffle = bool(config.get('shuffle_train', True))

    def __len__(self):
        return len(self.files) // self.bsz

    def batch(self, epoch, i):
        order = list(range(len(self.files)))
        if self.shuffle:
            g = torch.Generator().manual_seed(epoch)
            order = torch.randperm(len(order), generator=g).tolist()
        idx = order[i * self.bsz:(i + 1) * self.bsz]
        samples = []
        for j in idx:
            pc = torch.from_numpy(self.B.read_velodyne_bin(self.files[j][0]))
            pc = pc[torch.randperm(pc.shape[0])]                     # ShufflePointCloud (semantic_kitti_transforms.py:58-61)
            samples.append((pc, self.B.read_mask_cache(self.files[j][1])))
        return self.collate(samples)


def save_checkpoint(model, optimizer, path, epoch, metric_name, metric):
    torch.save({'state_dict': model.state_dict(), 'hyper_parameters': dict(getattr(model, 'hparams', {})),
                'optimizer_states': [optimizer.state_dict()], 'epoch': epoch, metric_name: metric}, path)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', '-c', type=str, help='Config file for current run', required=True)
    parser.add_argument('--train', '-t', action='store_true', help='Train the model')
    parser.add_argument('--test', '-e', action='store_true', help='Test the model')
    # extensions of this launcher
    parser.add_argument('--synthetic', action='store_true', help='synthetic scans instead of a dataset on disk')
    parser.add_argument('--data-root', default='data/SemanticKITTI')
    parser.add_argument('--max-epochs', type=int, default=1000)
    parser.add_argument('--max-steps', type=int, default=-1, help='stop after this many optimizer steps (-1: no limit)')
    parser.add_argument('--steps-per-epoch', type=int, default=100, help='synthetic data: batches per epoch')
    parser.add_argument('--compute-dtype', default=None, choices=[None, 'fp32', 'bf16', 'fp16'])
    parser.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (no HIP-graph replay)')
    parser.add_argument('--checkpoint-root', default='checkpoints')
    args = parser.parse_args(argv)

    is_training, is_testing = args.train, args.test
    if is_training is False and is_testing is False:
        is_training = True

    config_path = pathlib.Path(args.config)
    exp_name = config_path.stem
    checkpoint_folder_path = pathlib.Path(args.checkpoint_root).joinpath(exp_name)
    if not config_path.exists():
        raise ValueError(f'Could not find config at path {config_path}')
    with open(config_path, 'r') as f:
        config: dict = yaml.safe_load(f)
    if args.compute_dtype:
        config['compute_dtype'] = args.compute_dtype

    limit_val_batches = config.get('limit_val_batches', 1.0)
    check_metric = 'val_loss' if (limit_val_batches > 0 and not args.synthetic
                                  and config.get('dataset', 'semantic-kitti') != 'synthetic') else 'train_loss'
    if is_testing:
        regex = re.compile(r'(?:val|train)_loss=([0-9]+\.?[0-9]*)')
        checkpoints = [f for f in checkpoint_folder_path.iterdir() if not f.name.startswith('last')]
        best_checkpoint = min(checkpoints, key=lambda x: get_metric_from_name(x, regex))
        print(f'Testing from {best_checkpoint}')
        config['checkpoint'] = str(best_checkpoint)
        config['batch_size'] = config.get('test_batch_size', config.get('batch_size', 1))
        config['num_workers'] = config.get('test_num_workers', config.get('num_workers', 0))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('train_mask_bev_amd.py needs an MI355X (the product path has no CPU fallback)')
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('MBV_DIST_BACKEND', 'nccl')       # 'nccl' is RCCL on ROCm
        dist.init_process_group(backend, **({'device_id': device} if backend == 'nccl' else {}))

    model = MaskBevModule.from_config(config, checkpoint_folder_path).to(device)
    model.log_scalars = False
    model.flatten_parameters()         # fp16 compute: this also creates the device-side loss scaler
    opt_cfg = model.configure_optimizers()
    optimizer, scheduler = opt_cfg['optimizer'], opt_cfg['lr_scheduler']

    reducer = None
    if world > 1:
        from mask_bev_amd.ddp import GradientAllReducer
        reducer = GradientAllReducer(model, bucket_mb=64.0)

    dataset_name = 'synthetic' if args.synthetic else config.get('dataset', 'semantic-kitti')
    if dataset_name == 'synthetic':
        data = SyntheticBatches(config, device, rank, args.steps_per_epoch)
        val = None
    elif dataset_name == 'semantic-kitti':
        data = SemanticKittiCacheBatches(config, device, rank, world, args.data_root,
                                         config.get('train_sequences', [0, 1, 2, 3, 4, 5, 6, 7, 9, 10]))
        val = SemanticKittiCacheBatches(dict(config, shuffle_train=False), device, rank, world, args.data_root,
                                        config.get('val_sequences', [8])) if limit_val_batches > 0 else None
    else:
        raise NotImplementedError(dataset_name)

    def validate(epoch):
        if val is None:
            return None
        model.eval()
        tot, n = 0.0, 0
        with torch.no_grad():
            for i in range(len(val)):
                tot += float(model.validation_step(val.batch(epoch, i), i))
                n += 1
        model.train()
        v = tot / max(1, n)
        if world > 1:                  # every rank validates its own shard: the monitored value is the mean over ranks
            from mask_bev_amd.ddp import reduce_scalars
            v = reduce_scalars({'val_loss': torch.tensor(v, device=device)})['val_loss']
        return float(v)

    if is_training:
        model.train()
        if rank == 0:
            checkpoint_folder_path.mkdir(parents=True, exist_ok=True)
        graphed, step_count, best = None, 0, float('inf')
        # EarlyStopping(check_metric, patience=30) of the reference's Trainer (train_mask_bev.py:100): stop after 30
        # epochs without an improvement of the monitored metric (Lightning's default min_delta = 0, mode = 'min')
        es_best, es_wait, es_patience = float('inf'), 0, 30
        for epoch in range(args.max_epochs):
            model.current_epoch = epoch
            t0, loss_sum, n_steps = time.perf_counter(), None, 0
            n_batches = int(len(data) * float(config.get('limit_train_batches', 1.0))) if isinstance(
                config.get('limit_train_batches', 1.0), float) else int(config.get('limit_train_batches'))
            for i in range(max(1, n_batches)):
                batch = data.batch(epoch, i)
                if not args.no_graph and getattr(model, '_arena', None) is not None:
                    if graphed is None:
                        from mask_bev_amd.graph import GraphedTrainStep
                        if reducer is not None:
                            reducer.no_sync(True)
                        graphed = GraphedTrainStep(model, optimizer, batch, reducer=reducer)
                    loss = graphed.step(batch)
                else:
                    if reducer is not None:
                        reducer.sync_buffers()
                    loss = model.training_step(batch, i)
                    model.scale_loss(loss).backward()
                    if reducer is not None:
                        reducer.finish(optimizer)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=False)
                # accumulate on the device: the graphed step returns ONE static tensor that every replay overwrites,
                # so keeping references would average N aliases of the last batch's loss (the reference monitors the
                # epoch mean of train_loss, mask_bev_module.py:296 `self.log('train_loss', ..., on_epoch=True)`)
                loss_sum = loss.detach().clone() if loss_sum is None else loss_sum + loss.detach()
                n_steps += 1
                step_count += 1
                if 0 < args.max_steps <= step_count:
                    break
            train_loss = float(loss_sum) / max(1, n_steps)
            if world > 1:
                from mask_bev_amd.ddp import reduce_scalars
                train_loss = reduce_scalars({'train_loss': torch.tensor(train_loss, device=device)})['train_loss']
            val_loss = validate(epoch)
            monitored = val_loss if (check_metric == 'val_loss' and val_loss is not None) else train_loss
            if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                scheduler.step(train_loss)                          # monitor 'train_loss', interval 'epoch'
            else:
                scheduler.step()
            if rank == 0:
                dt = time.perf_counter() - t0
                print(f'epoch {epoch}: train_loss {train_loss:.6f}'
                      + (f' val_loss {val_loss:.6f}' if val_loss is not None else '')
                      + f'  {n_steps * int(config.get("batch_size", 1)) * world / dt:.1f} scans/s', flush=True)
                save_checkpoint(model, optimizer, checkpoint_folder_path / 'last.ckpt', epoch, check_metric, monitored)
                if monitored < best:
                    best = monitored
                    for old in checkpoint_folder_path.glob(f'{exp_name}-epoch=*'):
                        old.unlink()
                    save_checkpoint(model, optimizer, checkpoint_folder_path /
                                    f'{exp_name}-epoch={epoch:02d}-{check_metric}={monitored:.6f}.ckpt', epoch,
                                    check_metric, monitored)
            if monitored < es_best:
                es_best, es_wait = monitored, 0
            else:
                es_wait += 1
            if es_wait >= es_patience:         # `monitored` is rank-reduced, so every rank stops at the same epoch
                if rank == 0:
                    print(f'early stopping: {check_metric} did not improve in {es_patience} epochs '
                          f'(best {es_best:.6f})', flush=True)
                break
            if 0 < args.max_steps <= step_count:
                break
        if graphed is not None:
            graphed.close()

    if is_testing:
        # the reference runs trainer.validate then trainer.test (train_mask_bev.py:118-119); MaskBevModule defines no
        # test_step, so the test pass has nothing of its own to run: validation is the whole of it
        v = validate(0)
        if rank == 0:
            print(f'val_loss {v}' if v is not None else 'no validation data configured')

    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
