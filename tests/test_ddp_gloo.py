"""World-size-2 gloo test of the data-parallel path: bucketed, backward-overlapped gradient averaging gives the
same parameters on both ranks and the same update as a single process seeing the whole batch."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    """A rendezvous FILE, not a TCP port: a port found free here can be taken (or still sit in TIME_WAIT) by the time the
    ranks bind it — seen once as a failed rendezvous of the third test of a run; gloo picks its own pair ports."""
    fd, path = tempfile.mkstemp(prefix='mbv_gloo_rdv_')
    os.close(fd)
    os.unlink(path)                       # the FileStore creates it; a stale file of an earlier run would confuse it
    _RDV_FILES.append(path)
    return path


_RDV_FILES = []


@pytest.fixture(autouse=True)
def _remove_rendezvous_files():
    yield
    while _RDV_FILES:
        try:
            os.unlink(_RDV_FILES.pop())
        except OSError:
            pass


def _init_gloo(rank, world, rdv):
    dist.init_process_group('gloo', init_method=f'file://{rdv}', rank=rank, world_size=world)


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(8, 32), nn.ReLU(), nn.BatchNorm1d(32), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 1))


def _worker(rank, world, port, out):
    _init_gloo(rank, world, port)
    from mask_bev_amd.ddp import GradientAllReducer, reduce_scalars, shard_scans
    torch.manual_seed(100 + rank)            # different init per rank: construction must broadcast rank 0's
    m = nn.Sequential(nn.Linear(8, 32), nn.ReLU(), nn.BatchNorm1d(32), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 1))
    red = GradientAllReducer(m, bucket_mb=0.001)           # tiny buckets → several collectives in flight
    assert len(red.buckets) > 2
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    idx = shard_scans(range(8), rank, world)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for _ in range(3):
        red.sync_buffers()
        loss = ((m(x_all[idx]) - y_all[idx]) ** 2).mean()
        loss.backward()
        red.finish()
        opt.step()
        opt.zero_grad()
    scal = reduce_scalars({'loss': loss.detach(), 'rank': torch.tensor(float(rank))})
    out[rank] = ([p.detach().clone() for p in m.parameters()], scal)
    dist.destroy_process_group()


def test_gradient_allreduce_world2_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    p0, s0 = out[0]
    p1, s1 = out[1]
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                           # replicas stay identical
    assert s0 == s1 and s0['rank'] == pytest.approx(0.5)
    # single-process reference: rank 0's initial weights; mean of per-shard gradients == DDP semantics
    torch.manual_seed(100)
    ref = nn.Sequential(nn.Linear(8, 32), nn.ReLU(), nn.BatchNorm1d(32), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 1))
    import copy
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    shards = [list(range(r, 8, 2)) for r in range(2)]
    for _ in range(3):
        grads = []
        bn_state = copy.deepcopy(ref[2].state_dict())
        for r, idx in enumerate(shards):
            ref[2].load_state_dict(bn_state)               # every rank starts the step from rank 0's buffers
            loss = ((ref(x_all[idx]) - y_all[idx]) ** 2).mean()
            grads.append(torch.autograd.grad(loss, list(ref.parameters())))
            if r == 0:
                bn_after_rank0 = copy.deepcopy(ref[2].state_dict())
        ref[2].load_state_dict(bn_after_rank0)
        for p, g0, g1 in zip(ref.parameters(), *grads):
            p.grad = (g0 + g1) / 2
        opt.step()
        opt.zero_grad()
    for a, b in zip(p0, ref.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def _arena_worker(rank, world, port, out):
    _init_gloo(rank, world, port)
    from mask_bev_amd.arena import ParameterArena
    from mask_bev_amd.ddp import GradientAllReducer
    torch.manual_seed(100 + rank)
    enc, bb, head = nn.Linear(8, 32), nn.Linear(32, 16), nn.Linear(16, 1)
    m = nn.Sequential(enc, nn.ReLU(), bb, nn.ReLU(), head)
    red = GradientAllReducer(m, bucket_mb=0.001)           # broadcasts rank 0's parameters
    red.no_sync(True)                                      # graph-step mode: no hook-driven buckets
    arena = ParameterArena([('encoder', enc), ('backbone', bb), ('head', head)], shadow_dtype=None)
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    idx = list(range(rank, 8, world))
    for _ in range(2):
        ((m(x_all[idx]) - y_all[idx]) ** 2).mean().backward()          # accumulates into the arena views
        handles = red.start_arena(arena, ('head', 'backbone'), chunk_mb=0.0005)   # several chunks per segment
        handles += red.start_arena(arena, ('encoder',), chunk_mb=0.0005)
        red.finish_arena(arena, handles)                                # no optimizer hook → divides in place
        with torch.no_grad():
            arena.param.sub_(0.1 * arena.grad)
        arena.zero_grad()
    out[rank] = arena.param.clone()
    dist.destroy_process_group()


def test_arena_allreduce_world2_matches_single_process():
    """The graph step's data-parallel exchange: contiguous chunks of the arena gradient, no bucket copies."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_arena_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])
    torch.manual_seed(100)
    enc, bb, head = nn.Linear(8, 32), nn.Linear(32, 16), nn.Linear(16, 1)
    ref = nn.Sequential(enc, nn.ReLU(), bb, nn.ReLU(), head)
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    for _ in range(2):
        gs = []
        for r in range(2):
            idx = list(range(r, 8, 2))
            gs.append(torch.autograd.grad(((ref(x_all[idx]) - y_all[idx]) ** 2).mean(), list(ref.parameters())))
        with torch.no_grad():
            for p, g0, g1 in zip(ref.parameters(), *gs):
                p.sub_(0.1 * (g0 + g1) / 2)
    from mask_bev_amd.arena import ParameterArena
    want = ParameterArena([('encoder', enc), ('backbone', bb), ('head', head)], shadow_dtype=None).param
    assert torch.allclose(out[0], want, rtol=1e-5, atol=1e-6)


def _wire_worker(rank, world, port, out):
    _init_gloo(rank, world, port)
    from mask_bev_amd.arena import ParameterArena
    from mask_bev_amd.ddp import GradientAllReducer
    res = {}
    for wire in (None, torch.bfloat16):
        torch.manual_seed(100 + rank)
        enc, bb, head = nn.Linear(8, 32), nn.Linear(32, 16), nn.Linear(16, 1)
        m = nn.Sequential(enc, nn.ReLU(), bb, nn.ReLU(), head)
        red = GradientAllReducer(m, bucket_mb=0.001, grad_dtype=wire)     # broadcasts rank 0's parameters: same start both times
        red.no_sync(True)
        arena = ParameterArena([('encoder', enc), ('backbone', bb), ('head', head)], shadow_dtype=None)
        torch.manual_seed(7)
        x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
        idx = list(range(rank, 8, world))
        arena.zero_grad()
        ((m(x_all[idx]) - y_all[idx]) ** 2).mean().backward()
        local = arena.grad.clone()
        handles = red.start_ranges(arena, [arena.segments['head'], arena.segments['backbone']], chunk_mb=0.0005)
        handles += red.start_arena(arena, ('encoder',), chunk_mb=0.0005)
        red.finish_arena(arena, handles)
        res['f32' if wire is None else 'bf16'] = arena.grad.clone()
        res['local'] = local
    out[rank] = res
    dist.destroy_process_group()


def test_arena_allreduce_in_a_16_bit_wire_type():
    """``grad_dtype=torch.bfloat16``: the arena chunks of the graph step travel as bf16 (half the link bytes) and come back
    into the f32 gradient: identical on both ranks, equal to the bf16 sum of the bf16-rounded local gradients, within bf16
    rounding of the exact mean; the default (None) stays the exact in-place f32 exchange."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wire_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a['bf16'], b['bf16']) and torch.equal(a['f32'], b['f32'])
    exact = (a['local'] + b['local']) / 2
    assert torch.allclose(a['f32'], exact, rtol=1e-6, atol=1e-7)
    want = ((a['local'].bfloat16() + b['local'].bfloat16()).float()) / 2
    assert torch.equal(a['bf16'], want)
    assert (a['bf16'] - exact).abs().max() <= 2.0 ** -7 * exact.abs().max()
    assert not torch.equal(a['bf16'], a['f32'])


def _ranges_worker(rank, world, port, out):
    """The staged exchange of graph.py: sub-module ranges reduced as their gradients complete, one of them from a
    post-accumulate hook fired inside backward."""
    _init_gloo(rank, world, port)
    from mask_bev_amd.arena import ParameterArena
    from mask_bev_amd.ddp import GradientAllReducer
    torch.manual_seed(100 + rank)
    enc = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 32))
    bb = nn.Sequential(nn.Linear(32, 24), nn.ReLU(), nn.Linear(24, 16))
    head = nn.Linear(16, 1)
    m = nn.Sequential(enc, nn.ReLU(), bb, nn.ReLU(), head)
    red = GradientAllReducer(m, bucket_mb=0.001)
    red.no_sync(True)
    arena = ParameterArena([('encoder', enc), ('backbone', bb), ('head', head)], shadow_dtype=None)
    last = arena.range_of(bb[2])                            # "last stage" of the backbone
    a, b = arena.segments['backbone']
    assert a <= last[0] < last[1] == b and arena.range_of(bb) == (a, b)
    first_enc = arena.range_of(enc[2])                      # the encoder layer whose backward runs first
    ea, eb = arena.segments['encoder']
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    idx = list(range(rank, 8, world))
    from mask_bev_amd.ddp import RangeReady
    for _ in range(2):
        early = []
        # the range holds enc[2]'s weight AND bias: it goes on the wire when both have accumulated (RangeReady), not from
        # the bias's hook alone (autograd promises no order between a layer's parameters: that form failed 1 run in 8)
        guard = RangeReady(list(enc[2].parameters()), lambda: early.extend(red.start_ranges(arena, [first_enc]))).arm()
        red.debug_pending = [(guard, first_enc)]
        ((m(x_all[idx]) - y_all[idx]) ** 2).mean().backward()
        guard.remove()
        red.debug_pending = None
        assert early and guard.fired and not guard.pending(), 'the range was not launched from inside backward'
        handles = red.start_ranges(arena, [arena.segments['head'], last], chunk_mb=0.0005)
        handles += red.start_ranges(arena, [(a, last[0]), (last[1], b)])
        handles += early + red.start_ranges(arena, [(ea, first_enc[0]), (first_enc[1], eb)])
        red.finish_arena(arena, handles)
        with torch.no_grad():
            arena.param.sub_(0.1 * arena.grad)
        arena.zero_grad()
    out[rank] = arena.param.clone()
    dist.destroy_process_group()


def test_staged_range_allreduce_world2_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ranges_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])
    torch.manual_seed(100)
    enc = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 32))
    bb = nn.Sequential(nn.Linear(32, 24), nn.ReLU(), nn.Linear(24, 16))
    head = nn.Linear(16, 1)
    ref = nn.Sequential(enc, nn.ReLU(), bb, nn.ReLU(), head)
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 8), torch.randn(8, 1)
    for _ in range(2):
        gs = []
        for r in range(2):
            idx = list(range(r, 8, 2))
            gs.append(torch.autograd.grad(((ref(x_all[idx]) - y_all[idx]) ** 2).mean(), list(ref.parameters())))
        with torch.no_grad():
            for p, g0, g1 in zip(ref.parameters(), *gs):
                p.sub_(0.1 * (g0 + g1) / 2)
    from mask_bev_amd.arena import ParameterArena
    want = ParameterArena([('encoder', enc), ('backbone', bb), ('head', head)], shadow_dtype=None).param
    assert torch.allclose(out[0], want, rtol=1e-5, atol=1e-6)


def test_range_ready_fires_once_when_every_parameter_has_announced():
    """RangeReady: the callback runs exactly once per armed pass, only after the LAST parameter of the range announced
    (whatever the order, repeated announcements counted once), and `start_ranges`' debug check refuses a range whose
    guard still waits."""
    from mask_bev_amd.ddp import RangeReady
    lin = nn.Linear(4, 3)
    fired = []
    guard = RangeReady(list(lin.parameters()), lambda: fired.append(len(guard.pending()))).arm()
    for first, second in ((lin.bias, lin.weight), (lin.weight, lin.bias)):
        guard.arm()
        fired.clear()
        guard._announce(first)
        guard._announce(first)                      # announced twice (K3's backward does): still one parameter
        assert not fired and guard.pending() == [second] and not guard.fired
        guard._announce(second)
        guard._announce(second)
        assert fired == [0] and guard.fired
    guard.arm()
    fired.clear()
    lin(torch.randn(5, 4)).sum().backward()         # through autograd's own post-accumulate hooks
    assert fired == [0]
    guard.remove()

    class _Reducer:                                  # start_ranges' debug check, without a process group
        debug_pending = None
        grad_dtype = None
        from mask_bev_amd.ddp import GradientAllReducer as _G
        start_ranges = _G.start_ranges

        def _reduce_chunk(self, arena, lo, hi):
            return (lo, hi)

    red = _Reducer()
    guard = RangeReady(list(lin.parameters()), lambda: None).arm()
    red.debug_pending = [(guard, (10, 20))]
    assert red.start_ranges(None, [(0, 10), (20, 30)]) == [(0, 10), (20, 30)]          # disjoint: fine
    with pytest.raises(RuntimeError):
        red.start_ranges(None, [(5, 12)])
    guard._announce(lin.weight)
    with pytest.raises(RuntimeError):
        red.start_ranges(None, [(10, 20)])
    guard._announce(lin.bias)
    assert red.start_ranges(None, [(10, 20)]) == [(10, 20)]
    guard.remove()


def test_arena_range_of_rejects_interleaved_parameters():
    from mask_bev_amd.arena import ParameterArena
    a, b, c = nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4)
    arena = ParameterArena([('s', nn.Sequential(a, b, c))], shadow_dtype=None)
    assert arena.range_of(b)[0] == arena.range_of(a)[1]
    with pytest.raises(ValueError):
        arena.range_of([a.weight, c.weight])
    with pytest.raises(ValueError):
        arena.range_of(nn.Linear(2, 2))


def _worker_unused(rank, world, port, out):
    """Ranks with DIFFERENT sets of unused parameters: rank 1 never uses the `extra` branch.  The buckets must still be
    all-reduced in the same order on both ranks (index order), or the collectives pair up wrongly / hang."""
    _init_gloo(rank, world, port)
    from mask_bev_amd.ddp import GradientAllReducer

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.extra, self.b = nn.Linear(6, 12), nn.Linear(12, 12), nn.Linear(12, 1)

        def forward(self, x, use_extra):
            h = torch.relu(self.a(x))
            if use_extra:
                h = h + self.extra(h)
            return self.b(h)

    torch.manual_seed(3)
    m = Net()
    red = GradientAllReducer(m, bucket_mb=0.0001)          # (nearly) one bucket per parameter
    assert len(red.buckets) >= 4
    torch.manual_seed(11 + rank)
    x, y = torch.randn(5, 6), torch.randn(5, 1)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    for _ in range(2):
        ((m(x, use_extra=(rank == 0)) - y) ** 2).mean().backward()
        # a parameter announced twice (gradient accumulation outside autograd does that) must not launch early
        red._on_grad_ready(m.b.weight)
        red.finish()
        opt.step()
        opt.zero_grad()
    out[rank] = [p.detach().clone() for p in m.parameters()]
    dist.destroy_process_group()


def test_ranks_with_different_unused_parameters_stay_in_step():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_unused, args=(world, port, out), nprocs=world, join=True)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert out[0][2].abs().sum() > 0                       # the `extra` weight exists and moved (rank 0's half gradient)


def test_graph_step_reduce_plan_covers_the_arena_exactly_once():
    """The four pieces GraphedTrainStep launches (head + last stage, earlier stages + patch projection, the encoder's
    LayerNorm affine, the pillar feature net — graph.arena_reduce_plan) are disjoint, in that order, and cover every
    element of the arena gradient exactly once; each parameter lies wholly inside one piece.  CPU: the plan is pure
    arithmetic on the arena layout (the RCCL launches themselves: tests/test_ddp_graph_gpu.py, gloo on a GPU)."""
    from mask_bev_amd.graph import arena_reduce_plan
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from tests.util_cfg import tiny_kwargs
    m = MaskBevModule(**tiny_kwargs(nx=40, ny=40, q=4))
    from mask_bev_amd.arena import ParameterArena
    arena = ParameterArena([('encoder', m._encoder), ('backbone', m._backbone), ('head', m._panoptic_head)],
                           shadow_dtype=None)
    plan = arena_reduce_plan(m, arena)
    assert [name for name, _ in plan] == ['head + last backbone stage', 'earlier backbone stages + patch projection',
                                          'encoder LayerNorm affine', 'pillar feature net']
    cover = torch.zeros(arena.numel, dtype=torch.int32)
    for _, ranges in plan:
        for a, b in ranges:
            assert 0 <= a < b <= arena.numel
            cover[a:b] += 1
    assert bool((cover == 1).all())
    piece_of = {}
    for i, (_, ranges) in enumerate(plan):
        for a, b in ranges:
            for p, off in arena.layout:
                if a <= off and off + p.numel() <= b:
                    assert id(p) not in piece_of
                    piece_of[id(p)] = i
    assert len(piece_of) == len(arena.layout)                      # no parameter straddles two pieces
    names = {id(p): n for n, p in m.named_parameters()}
    by_piece = {i: {names[k] for k, v in piece_of.items() if v == i} for i in range(4)}
    assert all(n.startswith('_panoptic_head.') or '.stages.3.' in n or n.startswith('_backbone._backbone.norm3')
               for n in by_piece[0]), sorted(by_piece[0])[:5]
    assert by_piece[2] == {'_encoder._layer_norm.weight', '_encoder._layer_norm.bias'}
    assert all(n.startswith('_encoder._voxel_encoder.') or n.startswith('_encoder._pos_encoder.') for n in by_piece[3])
    assert all(n.startswith('_backbone.') for n in by_piece[1])
