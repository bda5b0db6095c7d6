"""RCCL executes the graph step's gradient exchange: the same `GraphedTrainStep(reducer=...)` path that
`bench.py --gpus N` takes, with `init_process_group('nccl', world_size=1, device_id=...)` — the one slice of the multi-GPU
row a single-GPU box can prove on hardware.  At world size 1 a SUM all-reduce is the identity, so everything RCCL touches
must come back unchanged, and what is under test is the ORDERING: RCCL's stream against the two graph replays, the in-place
all-reduce on arena views (four pieces: head + last stage between the replays, the earlier stages after the second, the
LayerNorm affine from its `RangeReady` gradient hook inside the eager encoder backward, the pillar feature net last), the
asynchronous buffer broadcast, `finish_arena` and `grad_scale` inside k_adamw.

Checked, exactly: the arena gradient the optimizer consumes equals, element for element, a snapshot of every range taken
ON THE COMPUTE STREAM right before that range's all-reduce was launched (no host synchronisation in between: the ordering
is the product's) — RCCL handed every byte back, and no kernel of graph 2 / the encoder backward wrote into a range that
was already on the wire; every range went out exactly once per step in plan order; `grad_scale == 1`; no hang (the child
is joined with a timeout).  Against the reducer-less run of the same seeds: the first loss is bit-equal and three steps
stay within 2e-3 — the tiny model's step is not bit-reproducible from process to process with or without the reducer
(scratch/rccl_determinism.py: 5e-4 in the parameters between two reducer-less runs), so that comparison cannot be exact.
With the bf16 wire the run stays within bf16 rounding.
Reference: /root/reference: train_mask_bev.py:92-96 (Lightning `strategy='ddp'`), SURVEY.md §8e."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _child(mode, rendezvous, out_file, compute_dtype):
    """mode: 'none' (no process group, no reducer) | 'f32' | 'bf16' (nccl world size 1, that wire type)."""
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from mask_bev_amd.graph import GraphedTrainStep, arena_reduce_plan
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from tests.util_cfg import random_gt, random_scans, tiny_kwargs
    red = None
    if mode != 'none':
        import torch.distributed as dist
        os.environ.update(RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', init_method=f'file://{rendezvous}', rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == 'nccl'
    torch.manual_seed(100)
    kw = dict(tiny_kwargs(nx=96, ny=96, q=8), compute_dtype=compute_dtype)
    m = MaskBevModule(**kw).to(dev).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 1500
    arena = m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    if mode != 'none':
        from mask_bev_amd.ddp import GradientAllReducer
        red = GradientAllReducer(m, grad_dtype=torch.bfloat16 if mode == 'bf16' else None)
        red.no_sync(True)
    batches = []
    for s in range(3):
        scans = [x.to(dev) for x in random_scans(kw, [2500, 3000], seed=s)]
        labels, gt = random_gt(kw, 2, 3, seed=50 + s)
        batches.append((scans, (labels.to(dev), gt.to(dev))))
    g = GraphedTrainStep(m, opt, batches[0], reducer=red)
    g.trace = []
    calls, scales, identity, covered_once = [], [], [], []
    snap = torch.zeros_like(arena.grad)
    covered = torch.zeros(arena.numel, dtype=torch.int32, device=dev)
    seen = {}
    if red is not None:
        orig = red.start_ranges

        def start_ranges(ar, ranges, chunk_mb=256.0):
            calls[-1].append([tuple(r) for r in ranges])
            for a, b in ranges:                        # stream-ordered copies, NO host synchronisation
                snap[a:b].copy_(ar.grad[a:b])
                covered[a:b] += 1
            return orig(ar, ranges, chunk_mb)

        red.start_ranges = start_ranges
        orig_step = opt.step

        def step():
            scales.append(float(opt.grad_scale))
            seen['reduced'] = arena.grad.clone()       # what k_adamw is about to consume (behind every h.wait())
            return orig_step()

        opt.step = step
    losses = []
    for i in range(3):
        calls.append([])
        covered.zero_()
        losses.append(float(g.step(batches[i])))
        torch.cuda.synchronize()
        if red is not None:
            identity.append(bool(torch.equal(seen['reduced'], snap)) and float(snap.abs().sum()) > 0)
            covered_once.append(bool((covered == 1).all()))
    torch.cuda.synchronize()
    plan = [[tuple(r) for r in rs] for _, rs in arena_reduce_plan(m, arena)]
    marks = [[name for name, _, _ in step_marks] for step_marks in g.trace]
    nbytes = [sum(nb for _, _, nb in step_marks) for step_marks in g.trace]
    torch.save(dict(params=arena.param.detach().cpu().clone(), losses=losses, calls=calls, plan=plan, scales=scales,
                    identity=identity, covered_once=covered_once,
                    marks=marks, nbytes=nbytes, numel=arena.numel,
                    buffers=torch.cat([b.detach().float().reshape(-1).cpu() for b in m.buffers() if b.is_floating_point()])),
               out_file)
    g.close()
    if mode != 'none':
        import torch.distributed as dist
        dist.destroy_process_group()


def _run(tmp_path, mode, compute_dtype):
    import torch.multiprocessing as mp
    out = tmp_path / f'{mode}.pt'
    ctx = mp.get_context('spawn')
    p = ctx.Process(target=_child, args=(mode, str(tmp_path / f'rdv_{mode}'), str(out), compute_dtype))
    p.start()
    p.join(600)
    if p.is_alive():                                     # a hang IS the failure this test exists to catch
        p.kill()
        p.join()
        pytest.fail(f'graph step with the nccl reducer ({mode} wire) did not finish: RCCL / graph replay ordering hang')
    assert p.exitcode == 0, f'child ({mode}) exited with {p.exitcode}'
    return torch.load(out)


@pytest.mark.parametrize('compute_dtype', ['bf16', 'fp32'])
def test_rccl_world1_graph_step_equals_the_reducerless_step(tmp_path, compute_dtype):
    ref = _run(tmp_path, 'none', compute_dtype)
    got = _run(tmp_path, 'f32', compute_dtype)
    # every piece of the plan went on the wire once per step, in plan order (3 of them directly; the LayerNorm affine
    # from its hook, between pieces 2 and 4)
    for step_calls in got['calls']:
        flat = [r for call in step_calls for r in call]
        want = [r for piece in got['plan'] for r in piece]
        assert flat == want, (flat, want)
    assert got['scales'] == [1.0] * 3
    for marks, nb in zip(got['marks'], got['nbytes']):
        assert any('LayerNorm affine (from its gradient hooks)' in s for s in marks), marks
        assert nb == 4 * got['numel']                    # the whole arena gradient, once
    assert got['identity'] == [True] * 3, 'the gradient the optimizer saw != the snapshot taken when its range went on the wire'
    assert got['covered_once'] == [True] * 3
    assert got['losses'][0] == ref['losses'][0]          # same parameters, same batch: the forward is deterministic
    for a, b in zip(got['losses'], ref['losses']):
        assert abs(a - b) <= 2e-3 * abs(b), (got['losses'], ref['losses'])
    d = float((got['params'] - ref['params']).abs().max())
    assert d < 2e-3, f'max |diff| {d:.3e} after 3 steps'  # three AdamW steps of lr 1e-4: process-to-process noise is ~5e-4
    assert torch.allclose(got['buffers'], ref['buffers'], rtol=1e-3, atol=1e-5)


def test_rccl_world1_graph_step_bf16_wire(tmp_path):
    ref = _run(tmp_path, 'none', 'bf16')
    got = _run(tmp_path, 'bf16', 'bf16')
    assert got['scales'] == [1.0] * 3 and got['covered_once'] == [True] * 3
    assert all(l == l and abs(l) < 1e4 for l in got['losses'])
    # gradients rounded to bf16 on the wire: AdamW's normalised update moves a parameter by <= lr per step whatever the
    # gradient's scale, so three steps differ by at most a few lr (1e-4) — and must not be identical (the wire was used)
    d = (got['params'] - ref['params']).abs().max()
    assert 0 < float(d) < 1e-3, float(d)
