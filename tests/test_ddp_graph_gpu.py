"""The path `bench.py --gpus N` takes, under a test: two fresh ranks (gloo rendezvous, both on cuda:0), each with
MaskBevModule + parameter arena + GradientAllReducer + GraphedTrainStep(reducer=...) — two HIP graphs with the arena
ranges all-reduced between / after them, the LayerNorm-affine all-reduce launched from its gradient hook, the 1/world
factor inside k_adamw.  Checked per step, exactly (a + b is commutative, so the sums are bit-equal):
  * every element of the arena gradient that reaches the optimizer is local(rank 0) + local(rank 1), where local(r)
    is what rank r held just before each range's all-reduce was launched — i.e. the optimizer consumes the MEAN of the
    two ranks' gradients once `grad_scale = 1/2` is applied, and no range was skipped or reduced twice;
  * `grad_scale == 0.5` when k_adamw is launched;
  * the replicas hold bit-identical parameters after 3 steps (checksum spread 0) although they see different scans.
Reference: /root/reference: train_mask_bev.py:92-96 (Lightning `strategy='ddp'`), SURVEY.md §2b C1-C6."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, out_dir, compute_dtype='bf16'):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    # file rendezvous (``port`` is a path under the test's tmp_path): a TCP port found free in the parent can be gone by
    # the time the ranks bind it
    dist.init_process_group('gloo', init_method=f'file://{port}', rank=rank, world_size=world)
    from mask_bev_amd.ddp import GradientAllReducer
    from mask_bev_amd.graph import GraphedTrainStep
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from tests.util_cfg import random_gt, random_scans, tiny_kwargs
    torch.manual_seed(100 + rank)                       # different initial weights: construction must broadcast rank 0's
    kw = dict(tiny_kwargs(nx=96, ny=96, q=8), compute_dtype=compute_dtype)
    m = MaskBevModule(**kw).to(dev).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 1500
    arena = m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    red = GradientAllReducer(m)
    assert red.arena is arena and not red._active        # arena parameters: no hook-driven buckets
    batches = []
    for s in range(3):                                   # each rank sees its own scans
        scans = [x.to(dev) for x in random_scans(kw, [2500, 3000], seed=10 * rank + s)]
        labels, gt = random_gt(kw, 2, 3, seed=50 + 10 * rank + s)
        batches.append((scans, (labels.to(dev), gt.to(dev))))
    g = GraphedTrainStep(m, opt, batches[0], reducer=red)

    calls = []
    local = torch.zeros_like(arena.grad)
    covered = torch.zeros(arena.numel, dtype=torch.int32, device=dev)
    orig_start = red.start_ranges

    def start_ranges(ar, ranges, chunk_mb=256.0):
        torch.cuda.synchronize()                         # the backward that fills these ranges has finished
        calls.append(list(ranges))
        for a, b in ranges:
            local[a:b].copy_(ar.grad[a:b])
            covered[a:b] += 1
        return orig_start(ar, ranges, chunk_mb)

    red.start_ranges = start_ranges
    seen = {}
    orig_step = opt.step

    def step():
        torch.cuda.synchronize()
        seen['grad_scale'] = opt.grad_scale
        seen['reduced'] = arena.grad.clone()
        return orig_step()

    opt.step = step
    ok_sum, ok_cov, scales = [], [], []
    for i in range(3):
        local.zero_()
        covered.zero_()
        g.step(batches[i])
        torch.cuda.synchronize()
        mine = local.cpu()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        ok_sum.append(bool(torch.equal(seen['reduced'].cpu(), both[0] + both[1])))
        ok_cov.append(bool((covered == 1).all()))
        if not ok_cov[-1]:
            bad = torch.nonzero(covered != 1).flatten()
            off = int(bad[0])
            name = next((n for n, p in m.named_parameters()
                         if p.data_ptr() <= arena.param.data_ptr() + 4 * off < p.data_ptr() + 4 * max(64, p.numel())), '?')
            print(f'rank {rank} step {i} calls {calls[-8:]}', flush=True)
            print(f'rank {rank} step {i}: {int((covered == 0).sum())} elements never reduced, '
                  f'{int((covered > 1).sum())} reduced more than once; first at {off} ({name}), segments {arena.segments}',
                  flush=True)
        scales.append(float(seen['grad_scale']))
    checksum = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().cpu()
    flat = arena.param.detach().cpu().clone()
    buffers = torch.cat([b.detach().float().reshape(-1).cpu() for b in m.buffers() if b.is_floating_point()])
    torch.save(dict(ok_sum=ok_sum, ok_cov=ok_cov, scales=scales, checksum=checksum, params=flat, buffers=buffers,
                    grad_norm=float(seen['reduced'].norm())), os.path.join(out_dir, f'rank{rank}.pt'))
    g.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('compute_dtype', ['bf16', 'fp32'])
def test_two_rank_graph_step_reduces_the_arena(tmp_path, compute_dtype):
    """(fp32: the K20 path — absmax records made inside the capture, the static input map's registered record, the grouped
    weight gradients — under the same two-rank graph step.)"""
    import torch.multiprocessing as mp
    world, port = 2, str(tmp_path / 'rendezvous')
    mp.spawn(_rank, args=(world, port, str(tmp_path), compute_dtype), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for r in (r0, r1):
        assert r['ok_cov'] == [True] * 3, 'an arena range was not reduced exactly once'
        assert r['ok_sum'] == [True] * 3, 'reduced gradient != sum of the ranks\' local gradients'
        assert r['scales'] == [0.5] * 3
        assert r['grad_norm'] > 0
    assert float(r0['checksum']) == float(r1['checksum'])          # replica_param_checksum_spread == 0
    assert torch.equal(r0['params'], r1['params'])
    # the floating-point buffers (BatchNorm running statistics) are rank 0's on every rank when a step ends — the
    # asynchronous broadcast behind the encoder forward (graph.py); the ranks saw different scans, so without it they differ
    assert r0['buffers'].numel() > 0 and torch.equal(r0['buffers'], r1['buffers'])
    assert float(r0['buffers'].abs().sum()) > 0
