#!/usr/bin/env python3
"""Benchmark of the MaskBEV scan -> BEV -> mask forward+backward path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched through
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``, one rank per GPU over RCCL).
A *step* is one full training step of ``MaskBevModule`` on one batch of synthetic SemanticKITTI-shaped
scans already resident in HBM: voxelise → PFN → scatter+LN → Swin → pixel decoder → masked-attention decoder
→ Hungarian-matched loss → backward (gradient all-reduce overlapped for N > 1) → AdamW step.
Rank 0 prints ONE JSON line (metric of BASELINE.json: LiDAR scans/s fwd+bwd, whole job).

Extra objects on the line:
  roofline      HBM roofline of the dominant hand-written kernel, timed with HIP events recorded by the library
                on its launch stream inside the timed region.
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='semantic_kitti_512')
    ap.add_argument('--batch', type=int, default=4, help='scans per GPU per step (YAML batch_size)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp16'])
    ap.add_argument('--distribution', default='lidar', choices=['lidar', 'uniform'],
                    help="synthetic point distribution: 64-beam LiDAR-shaped scans (headline) or uniform x/y "
                         "(worst case for the pillar count)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-budget-s', type=float, default=90.0)
    ap.add_argument('--pool', type=int, default=2, help='distinct synthetic batches kept resident in HBM')
    ap.add_argument('--no-arena', action='store_true', help='per-tensor parameters and torch.optim.AdamW')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying the HIP graph')
    return ap.parse_args()


def cpu_baseline(workload: str, module, budget_s: float):
    """Oracle forward + loss + backward on the host cores, batch of 1 scan of the same workload, fp32.
    ORACLE USE: checker/baseline only — never on the measured GPU path.
    Threads: 6, the value the reference pins (``OMP_NUM_THREADS=6``, /root/reference: train_mask_bev.py:14);
    oversubscribing a many-core host with tiny torch ops is an order of magnitude slower."""
    from oracle import maskbev_oracle as O
    from mask_bev_amd import synthetic
    kw = synthetic.module_kwargs(workload, 1)
    cfg = O.make_cfg(**kw)
    sd = {k: v.detach().float().cpu() for k, v in module.state_dict().items()}
    sd_g = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running_' not in k else v.clone())
            for k, v in sd.items()}
    cores = min(6, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    scans, (labels, masks) = synthetic.make_batch(workload, 1, 0, 10_000, torch.device('cpu'))
    t0 = time.perf_counter()
    cls, mk, _ = O.model_forward(cfg, sd_g, scans, training=True)
    loss = O.total_loss(O.loss_dict(cfg, cls, mk, labels, masks, O.PointSource(0)))
    t_fwd = time.perf_counter() - t0
    if t_fwd > budget_s / 3.0:      # keep the default run within minutes: do not start the backward
        return dict(value=1.0 / (3.0 * t_fwd), unit='scans/s', cores=cores, kind='port',
                    sample=f'1 scan of {workload}, fp32, oracle forward+loss only ({t_fwd:.1f} s, over budget); '
                           f'value assumes backward = 2x forward; torch threads = {cores}')
    loss.backward()
    t_all = time.perf_counter() - t0
    return dict(value=1.0 / t_all, unit='scans/s', cores=cores, kind='port',
                sample=f'1 scan of {workload}, fp32, oracle forward+loss+backward, 1 iteration '
                       f'(forward+loss {t_fwd:.1f} s, total {t_all:.1f} s), torch threads = {cores} '
                       f'(the reference pins OMP_NUM_THREADS=6)')


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world == 1 and args.gpus > 1:
        # launched without torchrun: spawn it as a child and exit with its code
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)      # one rank per GPU; wraps only in the single-GPU smoke test of the N > 1 path
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('MBV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; gloo only for smoke tests
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from mask_bev_amd import ops, synthetic, tuning
    from mask_bev_amd.mask_bev_module import MaskBevModule

    torch.manual_seed(420)
    kw = synthetic.module_kwargs(args.workload, args.batch, compute_dtype=args.dtype)
    tuned = tuning.use_tuned_gemms()     # hipBLASLt solution table for this step's GEMM shapes (look-up only)
    model = MaskBevModule(**kw).to(device)
    model.train()
    model.log_scalars = False           # scalar logging is host-side bookkeeping, not the path
    if not args.no_arena:
        model.flatten_parameters()      # flat param / grad / bf16-shadow arena + single-launch AdamW (K11)
    opt = model.configure_optimizers()['optimizer']
    reducer = None
    if world > 1:
        from mask_bev_amd.ddp import GradientAllReducer
        reducer = GradientAllReducer(model, bucket_mb=64.0)
        if getattr(model, '_arena', None) is not None:
            model._arena.refresh_shadow()       # the construction-time parameter broadcast wrote the f32 arena
            # arena gradients are accumulated in place, once per USE of a parameter (a packed in_proj weight is used
            # by three projections): the per-parameter ready-hooks of the bucketed reducer do not apply — the arena
            # is reduced in contiguous ranges after (graph step: during) the backward
            reducer.no_sync(True)

    pool = [synthetic.make_batch(args.workload, args.batch, rank, s, device, args.distribution)
            for s in range(args.pool)]
    torch.cuda.synchronize()

    graphed = None
    if not args.no_graph:
        from mask_bev_amd.graph import GraphedTrainStep
        if reducer is not None:
            reducer.no_sync(True)          # gradients are averaged after the replay, not from hooks inside it
        graphed = GraphedTrainStep(model, opt, pool[0], reducer=reducer)

    def step(i: int):
        scans, gt = pool[i % len(pool)]
        if graphed is not None:
            return graphed.step((scans, gt))
        if reducer is not None:
            reducer.sync_buffers()
        loss = model.training_step((scans, gt), i)
        loss.backward()
        if reducer is not None:
            if getattr(model, '_arena', None) is not None:
                reducer.reduce_arena(model._arena, opt)
            else:
                reducer.finish()
        opt.step()
        opt.zero_grad(set_to_none=args.no_arena)
        return loss

    for i in range(args.warmup):
        step(i)
    ops.TIMER.reset()
    ops.TIMER.enabled = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.TIMER.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    replica_spread = None
    if world > 1:
        # data-parallel sanity, outside the timed region: every rank must hold the same parameters after the run
        cs = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replica_spread = float((hi - lo).item())

    if rank == 0:
        w = synthetic.WORKLOADS[args.workload]
        nx = int((w['x_range'][1] - w['x_range'][0]) / w['voxel_size'])
        ny = int((w['y_range'][1] - w['y_range'][0]) / w['voxel_size'])
        c = kw['encoder_feat_channels'][-1]
        times = ops.TIMER.summary_ms()
        # algorithmic HBM bytes per launch: fwd reads weight+bias once per batch and writes the
        # output; bwd reads grad_out and weight, writes grad_weight+grad_bias.  V*C*4 pillar rows are < 2 % and
        # left out (stated in DESIGN.md), as is the 4 B/cell map.
        cells = nx * ny
        acc = 0 if args.no_arena else 2 * c * cells          # arena: d(weight), d(bias) are read and accumulated into
        n_params = sum(p.numel() for p in model.parameters())
        # bf16 compute: the map leaves K3 (and its gradient comes back) as the backbone's bf16 patch rows
        io = 2.0 if model._patch_handoff() else 4.0
        algo = {'k_ln_apply': 2 * c * cells * 4.0 + args.batch * c * cells * io,
                'k_ln_bwd_dense': args.batch * c * cells * io + (c * cells + 2 * c * cells + acc) * 4.0,
                # K11: read param, grad, exp_avg, exp_avg_sq; write param, exp_avg, exp_avg_sq, zeroed grad, bf16 shadow
                'k_adamw': n_params * (16.0 + 16.0 + 2.0)}
        # HBM bytes per launch from the PMC passes of profiles/r01 (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction of
        # MI355X_MICROARCH.md); same workload, same build
        traffic = {'k_ln_apply': 568.8e6, 'k_ln_bwd_dense': 993.7e6, 'k_adamw': 6511.5e6} \
            if (args.workload == 'semantic_kitti_512' and args.batch == 4 and not args.no_arena and io == 2.0
                and args.distribution == 'lidar') else {}
        roof = {}
        for name, ms in times.items():
            if not ms or name not in algo:
                continue
            avg = sum(ms) / len(ms)
            roof[name] = dict(bound='hbm', achieved=algo[name] / (avg * 1e-3) / 1e9, peak=8000.0, unit='GB/s',
                              frac=algo[name] / (avg * 1e-3) / 1e9 / 8000.0, traffic=traffic.get(name),
                              kernel=name, avg_ms=avg, launches=len(ms), algorithmic_bytes=algo[name])
        dominant = max(roof.values(), key=lambda r: r['avg_ms']) if roof else None
        # whole-step figure: algorithmic work per scan of the S2 configuration — 0.97 TFLOP and 4.5 GB
        # forward + backward, plus the optimizer pass shared by the scans of a step — against the chip peaks
        step_roof = None
        if args.workload == 'semantic_kitti_512':
            sps = args.batch * args.steps / dt                        # scans per second on this GPU
            bytes_scan = 4.5e9 + algo['k_adamw'] / args.batch
            step_roof = dict(flops_per_scan=0.97e12, achieved_tflops=0.97 * sps, mfma_peak_tflops=2500.0,
                             frac_mfma=0.97 * sps / 2500.0, bytes_per_scan=bytes_scan,
                             achieved_gbs=bytes_scan * sps / 1e9, hbm_peak_gbs=8000.0,
                             frac_hbm=bytes_scan * sps / 1e9 / 8000.0)
        line = dict(
            metric='LiDAR scans/sec fwd+bwd, 120k-pt 512x512 BEV 100q', value=args.batch * world * args.steps / dt,
            unit='scans/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
            higher_is_better=True, scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic' if args.distribution == 'lidar' else 'synthetic (uniform x/y points)',
            config=dict(workload=f'{args.workload}: {w["points"]} pts/scan, {ny}x{nx} BEV, {w["num_queries"]} queries',
                        scans_per_gpu=args.batch, global_batch=args.batch * world, parallelism=f'dp{world}',
                        step='fwd + Hungarian loss + bwd + AdamW', launch='eager' if args.no_graph else 'hip-graph', tuned_gemm_table=tuned, replica_param_checksum_spread=replica_spread,
                        final_loss=final_loss),
            roofline=dominant, roofline_all=list(roof.values()), step_roofline=step_roof)
        if not args.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(args.workload, model, args.cpu_baseline_budget_s)
            except Exception as e:  # the baseline must never take the GPU number down with it
                line['cpu_baseline'] = dict(value=None, unit='scans/s', cores=os.cpu_count(), kind='port',
                                            sample=f'failed: {type(e).__name__}: {e}')
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
