"""fp16 compute (BASELINE.json configs[4]'s dtype; a build extension like bf16 — the reference trains fp32 only,
/root/reference: train_mask_bev.py:96 `precision=32`): the optimizer-side pieces that exist only for IEEE half.
  * FlatAdam with an fp16 weight shadow and the device-side LossScaler reproduces torch.optim.AdamW driven by
    torch.amp.GradScaler's rules — un-scaling inside k_adamw, skip + back-off on inf / nan, growth after a clean streak —
    without a host synchronisation in `step()`;
  * the half instantiations of K4 / K6 / K7 / K12 / K13 / K3 are covered by `dtype` parameters of those kernels' own
    tests, the whole model by tests/test_model_gpu.py::test_16bit_whole_model_against_fp32_oracle[fp16] and
    ::test_waymo_scale_fp16_trains_with_device_loss_scaling;
  * the HIP-graph step in fp16 replays the captured loss-scale multiply and tracks the eager fp16 step."""
import pytest
import torch

from tests.util_cfg import random_gt, random_scans, tiny_kwargs

pytestmark = pytest.mark.gpu


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(37, 64)
        self.n = torch.nn.LayerNorm(64)
        self.b = torch.nn.Linear(64, 5, bias=False)


def test_flat_adam_fp16_shadow_and_device_loss_scaler(device):
    from mask_bev_amd.arena import FlatAdam, LossScaler, ParameterArena
    torch.manual_seed(1)
    ref, mine = _Toy().to(device), _Toy().to(device)
    mine.load_state_dict(ref.state_dict())
    arena = ParameterArena([('all', mine)], shadow_dtype=torch.float16)
    scaler = LossScaler(device, init_scale=4096.0, growth_interval=3)
    opt = FlatAdam(arena, [dict(segment='all', lr=3e-3)], lr=3e-3, weight_decay=0.05, decoupled=True, scaler=scaler)
    topt = torch.optim.AdamW(ref.parameters(), lr=3e-3, weight_decay=0.05)
    g = torch.Generator(device='cpu').manual_seed(5)
    scale, streak = 4096.0, 0
    applied = 0
    for it in range(9):
        overflow = it in (2, 6)
        for p, q in zip(ref.parameters(), mine.parameters()):
            gr = torch.randn(p.shape, generator=g).to(device)
            p.grad = gr.clone()
            q.grad.copy_(gr * scale)                      # what a backward of loss * scale leaves in the arena
        if overflow:
            victim = list(mine.parameters())[it % 3]
            victim.grad.view(-1)[3] = float('inf') if it == 2 else float('nan')
        before = arena.param.clone()
        m_before = opt.exp_avg.clone()
        opt.step()
        if overflow:                                      # GradScaler.step skips, update() backs off
            assert torch.equal(arena.param, before) and torch.equal(opt.exp_avg, m_before)
            scale, streak = max(scale * 0.5, 1.0), 0
        else:
            # torch's reference step, untouched: a skipped step does not advance Adam's count in either (the device-side
            # count of applied updates, LossScaler.applied_steps, drives k_adamw's bias corrections)
            topt.step()
            applied += 1
            assert int(scaler.applied_steps.item()) == applied
            streak += 1
            if streak == 3:
                scale, streak = scale * 2.0, 0
            for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
                assert torch.allclose(p, q, rtol=3e-6, atol=3e-7), (it, n, float((p - q).abs().max()))
                assert torch.equal(q._mbv_shadow, q.detach().to(torch.float16))      # shadow = RNE half of the new value
        assert float(arena.grad.abs().max()) == 0.0                                   # cleared by the same pass
        assert scaler.get_scale() == scale, (it, scaler.get_scale(), scale)
        assert int(scaler.flag.item()) == 0
    assert applied == 7
    sd = opt.state_dict()
    assert sd['loss_scaler']['scale'] == scale
    assert sd['flat_state']['steps'] == 7 and sd['loss_scaler']['applied_steps'] == 7      # not the 9 calls


def test_graphed_fp16_step_tracks_eager(device):
    """GraphedTrainStep under compute_dtype='fp16': the captured step multiplies the loss by the device-side scale,
    FlatAdam un-scales; three replays give finite losses close to the eager fp16 losses of a twin module (same
    weights, batches; fresh random sampling points: 6 %), the parameters move, and the scale stays a power of two."""
    from mask_bev_amd.graph import GraphedTrainStep
    from mask_bev_amd.mask_bev_module import MaskBevModule
    kw = dict(tiny_kwargs(nx=96, ny=96, q=8), compute_dtype='fp16')
    torch.manual_seed(0)
    eager = MaskBevModule(**kw).to(device).train()
    torch.manual_seed(0)
    graphed = MaskBevModule(**kw).to(device).train()
    graphed.load_state_dict(eager.state_dict())
    batches = []
    for s in range(3):
        scans = [x.to(device) for x in random_scans(kw, [2500, 3000], seed=s)]
        labels, gt = random_gt(kw, 2, 3, seed=20 + s)
        batches.append((scans, (labels.to(device), gt.to(device))))
    for m in (eager, graphed):
        m.log_scalars = False
        m._panoptic_head._panoptic_head.num_points = 1500
    a_e, a_g = eager.flatten_parameters(), graphed.flatten_parameters()
    for m in (eager, graphed):             # start below the overflow point of this model (the default 2^16 backs off
        m._loss_scaler.scale.fill_(256.0)  # over the first few steps; that path is exercised by the other tests)
    o_e, o_g = eager.configure_optimizers()['optimizer'], graphed.configure_optimizers()['optimizer']
    g = GraphedTrainStep(graphed, o_g, batches[0])
    start = a_g.param.clone()
    side = torch.cuda.Stream()
    for i in range(3):
        lg = float(g.step(batches[i]))
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            le = eager.training_step(batches[i], i)
            eager.scale_loss(le).backward()
            o_e.step()
            le = float(le.detach())
        torch.cuda.current_stream().wait_stream(side)
        assert lg == lg and le == le and abs(lg - le) / le < 0.06, (i, lg, le)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a_g.param).all()) and not torch.equal(start, a_g.param)
    s = graphed._loss_scaler.get_scale()
    assert s in (64.0, 128.0, 256.0)
    g.close()
