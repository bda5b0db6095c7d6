"""Parameter arena: every parameter, gradient and optimizer moment of a ``MaskBevModule`` laid out in four flat
HBM buffers (288 GB of HBM3E makes the extra bf16 shadow free), so that the per-step bookkeeping of ≈ 700 tensors
collapses into a handful of launches (K11, csrc/optim.hip):

* ``param``   f32 — ``p.data`` of every parameter is a view into it (checkpoint keys and shapes unchanged);
* ``grad``    f32 — ``p.grad`` is a view; the Linear layers accumulate their weight / bias gradients straight into
                    it (``ops.linear``), autograd adds the rest in place; one fill clears it, and the gradient
                    all-reduce runs over contiguous chunks of it with no bucket copies;
* ``shadow``  bf16 — the copy of the weights the bf16 GEMMs read, written by the optimizer kernel itself instead
                    of ≈ 450 per-layer cast kernels per step;
* ``exp_avg``, ``exp_avg_sq`` f32 — Adam moments (owned by :class:`FlatAdam`).

The arena is laid out in the reference's differential-lr groups — encoder, backbone, head
(/root/reference: mask_bev/mask_bev_module.py:131-139) — each a contiguous segment, so an optimizer step is one
``mbv_adamw_step`` launch per group (one launch in total when the groups share a learning rate).

Build it after the module is on its GPU (``module.to(device)`` re-allocates parameters and would detach the views).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops
from ._lib import check

_ALIGN = 64          # elements: every parameter starts on a 256-byte boundary


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class ParameterArena:
    def __init__(self, segments: Iterable[Tuple[str, nn.Module]], shadow_dtype: Optional[torch.dtype] = torch.bfloat16):
        segments = list(segments)
        seen = set()
        self.layout: List[Tuple[nn.Parameter, int]] = []          # (parameter, offset)
        self.segments: Dict[str, Tuple[int, int]] = {}
        off = 0
        device = None
        for name, mod in segments:
            start = off
            for p in mod.parameters():
                if id(p) in seen:
                    continue
                seen.add(id(p))
                if p.dtype != torch.float32:
                    raise TypeError('the arena holds f32 master parameters')
                device = device or p.device
                if p.device != device:
                    raise ValueError('all parameters must live on one device')
                self.layout.append((p, off))
                off += _round_up(p.numel(), _ALIGN)
            self.segments[name] = (start, off)
        if device is None:
            raise ValueError('no parameters')
        self.numel = off
        self.device = device
        self.param = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = None
        if shadow_dtype is not None:
            if shadow_dtype not in (torch.bfloat16, torch.float16):
                raise TypeError('the shadow copy is bf16 or fp16')
            self.shadow = torch.zeros(off, dtype=shadow_dtype, device=device)
        with torch.no_grad():
            for p, o in self.layout:
                n = p.numel()
                view = self.param[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + n].view(p.shape)
                p._mbv_arena = True                         # ops.linear accumulates dW / db into p.grad directly
                if self.shadow is not None:
                    p._mbv_shadow = self.shadow[o:o + n].view(p.shape)
        self.refresh_shadow()

    # -- views ------------------------------------------------------------------------------------------
    def segment(self, name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        a, b = self.segments[name]
        return self.param[a:b], self.grad[a:b]

    def span(self, names: Iterable[str]) -> Tuple[int, int]:
        """Smallest contiguous range covering the named segments."""
        ab = [self.segments[n] for n in names]
        return min(a for a, _ in ab), max(b for _, b in ab)

    def range_of(self, module_or_params) -> Tuple[int, int]:
        """Arena range [lo, hi) holding exactly the parameters of a sub-module (they are laid out in registration
        order, so a sub-module's parameters are contiguous); raises if other parameters are interleaved."""
        params = list(module_or_params.parameters()) if isinstance(module_or_params, nn.Module) else list(module_or_params)
        ids = {id(p) for p in params}
        inside = [(o, o + _round_up(p.numel(), _ALIGN)) for p, o in self.layout if id(p) in ids]
        if len(inside) != len(ids):
            raise ValueError('parameters outside the arena')
        lo, hi = min(a for a, _ in inside), max(b for _, b in inside)
        if any(lo <= o < hi and id(p) not in ids for p, o in self.layout):
            raise ValueError('the parameters are not contiguous in the arena')
        return lo, hi

    # -- maintenance ------------------------------------------------------------------------------------
    @property
    def shadow_flag(self) -> int:
        """MBV_DT_BF16 / MBV_DT_F16 (maskbev_hip.h); 0 without a shadow."""
        return 0 if self.shadow is None else (2 if self.shadow.dtype == torch.float16 else 1)

    def refresh_shadow(self):
        """Re-derive the 16-bit shadow from the f32 parameters (after load_state_dict / broadcast / manual edits)."""
        ops.note_parameters_changed()
        if self.shadow is None or self.numel == 0:
            return
        if self.device.type != 'cuda':
            self.shadow.copy_(self.param)
            return
        lib = _lib.load()
        check(lib.mbv_refresh_shadow(self.param.data_ptr(), self.shadow.data_ptr(), self.shadow_flag, self.numel,
                                     torch.cuda.current_stream(self.device).cuda_stream), 'mbv_refresh_shadow')

    def zero_grad(self, names: Optional[Iterable[str]] = None):
        if names is None:
            self.grad.zero_()
        else:
            a, b = self.span(names)
            self.grad[a:b].zero_()

    def rebind(self):
        """Re-attach ``p.grad`` views (an ``optimizer.zero_grad(set_to_none=True)`` drops them)."""
        for p, o in self.layout:
            p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def intact(self) -> bool:
        base = self.param.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base for p, _ in self.layout)


class LossScaler:
    """Dynamic loss scaling for fp16 compute with ``torch.amp.GradScaler``'s rules (start at 2**16, halve on a
    non-finite gradient and skip that update, double after ``growth_interval`` clean steps) — but with the scale, the
    overflow flag and the clean-step count held on the device and read / updated by kernels (K11:
    ``mbv_grad_nonfinite``, ``mbv_adamw_step(loss_scale, skip_flag)``, ``mbv_loss_scale_update``), so that a step
    neither synchronises with the host nor changes shape: ``scale_loss`` is captured in the HIP graph like any other
    multiply.  Gradients are left scaled in the arena; the optimizer kernel divides on the fly."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.device = torch.device(device)
        self.scale = torch.full((1,), float(init_scale), dtype=torch.float32, device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.clean_steps = torch.zeros(1, dtype=torch.int32, device=self.device)
        # updates actually APPLIED (an overflowed step is skipped): Adam's bias-correction count, read by k_adamw
        self.applied_steps = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval = int(growth_interval)

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self.scale.view(())

    def check(self, grad: torch.Tensor):
        """flag |= any non-finite element of ``grad`` (the all-reduced arena gradient)."""
        lib = _lib.load()
        check(lib.mbv_grad_nonfinite(grad.data_ptr(), grad.numel(), self.flag.data_ptr(),
                                     torch.cuda.current_stream(self.device).cuda_stream), 'mbv_grad_nonfinite')

    def update(self):
        lib = _lib.load()
        check(lib.mbv_loss_scale_update(self.scale.data_ptr(), self.clean_steps.data_ptr(), self.flag.data_ptr(),
                                        self.growth_factor, self.backoff_factor, self.growth_interval,
                                        self.applied_steps.data_ptr(),
                                        torch.cuda.current_stream(self.device).cuda_stream), 'mbv_loss_scale_update')

    def get_scale(self) -> float:
        return float(self.scale.item())

    def state_dict(self):
        return dict(scale=self.get_scale(), clean_steps=int(self.clean_steps.item()),
                    applied_steps=int(self.applied_steps.item()),
                    growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                    growth_interval=self.growth_interval)

    def load_state_dict(self, sd):
        self.scale.fill_(float(sd['scale']))
        self.clean_steps.fill_(int(sd.get('clean_steps', 0)))
        if 'applied_steps' in sd:
            self.applied_steps.fill_(int(sd['applied_steps']))
        self.growth_factor = float(sd.get('growth_factor', self.growth_factor))
        self.backoff_factor = float(sd.get('backoff_factor', self.backoff_factor))
        self.growth_interval = int(sd.get('growth_interval', self.growth_interval))


class FlatAdam(torch.optim.Optimizer):
    """Adam / AdamW over a :class:`ParameterArena`: one ``mbv_adamw_step`` launch per parameter group.

    ``param_groups`` carry ``lr`` / ``weight_decay`` / ``betas`` / ``eps`` like torch's optimizers, so the
    reference's schedulers (ReduceLROnPlateau, CosineAnnealingLR — mask_bev_module.py:153-159) drive it unchanged.
    Adjacent groups with identical hyper-parameters are fused into one launch.  ``step()`` also refreshes the 16-bit
    shadow and clears the gradient in the same pass (``zero_grad=True``).  With a :class:`LossScaler` (fp16 compute) the
    gradient is checked for inf / nan first, un-scaled inside the update kernel, and an overflowed step leaves
    parameters and moments untouched, and Adam's bias-correction count does not advance either: with a scaler the count
    of APPLIED updates lives on the device (``LossScaler.applied_steps``, incremented by ``mbv_loss_scale_update``), as
    ``torch.amp.GradScaler`` skipping ``optimizer.step()`` would leave torch's.  ``state_dict()['flat_state']['steps']``
    is that count."""

    def __init__(self, arena: ParameterArena, groups: List[dict], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, decoupled: bool = True, zero_grad: bool = True,
                 scaler: Optional[LossScaler] = None):
        self.arena = arena
        self.scaler = scaler
        self.decoupled = decoupled
        self.zero_grad_in_step = zero_grad
        self.grad_scale = 1.0
        frozen = [tuple(p.shape) for p, _ in arena.layout if not p.requires_grad]
        if frozen:          # the single launch updates (and decays) every element of a segment; torch skips frozen ones
            raise ValueError(f'FlatAdam updates whole arena segments: {len(frozen)} frozen parameter(s) in the arena '
                             f'(first shape {frozen[0]}); keep frozen parameters out of the arena')
        pgs = []
        for g in groups:
            a, b = arena.segments[g['segment']]
            flat = nn.Parameter(arena.param[a:b], requires_grad=True)
            flat.grad = arena.grad[a:b]
            pg = {k: v for k, v in g.items() if k != 'segment'}
            pg['params'] = [flat]
            pg['segment'] = g['segment']
            pgs.append(pg)
        super().__init__(pgs, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(arena.param)
        self.exp_avg_sq = torch.zeros_like(arena.param)
        self.steps = 0
        self._k3 = None              # (weight, bias, offset_w, offset_b) of a LayerNorm whose update K3's backward performs
        self._k3_applied = False

    # -- K3 backward + AdamW of the (C, ny, nx) LayerNorm affine in one launch (csrc/scatter_layernorm.hip, ADAM) ------------
    def fuse_layernorm_affine(self, weight: Optional[nn.Parameter], bias: Optional[nn.Parameter] = None) -> bool:
        """Ask K3's backward to perform the AdamW update of ``weight`` / ``bias`` (the encoder's (C, ny, nx) LayerNorm affine,
        a third of the model's parameters) itself: their gradients are complete in that kernel's registers, so they never
        reach the arena — 16 B per parameter of gradient traffic less per step.  The CALLER guarantees one backward pass
        per ``step()`` (the graph step does); a second pass before ``step()`` raises.  Not with a loss scaler (an overflowed
        step must skip EVERY update, and the verdict is only known after the whole backward) and not with a gradient
        all-reduce (``grad_scale`` != 1: the reduced gradient does not exist inside the kernel).  ``None`` disarms.
        Returns whether the fusion is armed."""
        if ops.K3_ADAM[0] is self:
            ops.K3_ADAM[0] = None
        self._k3, self._k3_applied = None, False
        if weight is None or bias is None or self.scaler is not None or not self.zero_grad_in_step:
            return False
        off = {id(p): o for p, o in self.arena.layout}
        if id(weight) not in off or id(bias) not in off or weight.shape != bias.shape:
            return False
        self._k3 = (weight, bias, off[id(weight)], off[id(bias)])
        ops.K3_ADAM[0] = self
        return True

    def claim(self, weight, bias):
        """Called by K3's backward: the optimizer state of (weight, bias) for the fused update, or None (not the armed
        parameters / a data-parallel step).  Marks the update as applied for the coming ``step()``."""
        k3 = self._k3
        if k3 is None or weight is not k3[0] or bias is not k3[1] or float(self.grad_scale) != 1.0:
            return None
        if self._k3_applied:
            raise _lib.MaskBevHipError('FlatAdam: K3 backward ran twice before step() with the fused LayerNorm-affine update '
                                       'armed (gradient accumulation needs fuse_layernorm_affine(None))')
        _, _, ow, ob = k3
        pg = self._group_of(ow)
        n = weight.numel()
        if self._group_of(ob + n - 1) is not pg:
            return None
        ar = self.arena
        self._k3_applied = True
        return dict(m_w=self.exp_avg.data_ptr() + 4 * ow, v_w=self.exp_avg_sq.data_ptr() + 4 * ow,
                    m_b=self.exp_avg.data_ptr() + 4 * ob, v_b=self.exp_avg_sq.data_ptr() + 4 * ob,
                    sh_w=0 if ar.shadow is None else ar.shadow.data_ptr() + 2 * ow,
                    sh_b=0 if ar.shadow is None else ar.shadow.data_ptr() + 2 * ob, shadow_flag=ar.shadow_flag,
                    lr=float(pg['lr']), beta1=float(pg['betas'][0]), beta2=float(pg['betas'][1]), eps=float(pg['eps']),
                    weight_decay=float(pg['weight_decay']), step=self.steps + 1, decoupled=1 if self.decoupled else 0)

    def _group_of(self, offset: int):
        for pg in self.param_groups:
            a, b = self.arena.segments[pg['segment']]
            if a <= offset < b:
                return pg
        return None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ar = self.arena
        if ar.device.type != 'cuda':
            raise _lib.MaskBevHipError('FlatAdam runs on the MI355X only (mbv_adamw_step)')
        ops.note_parameters_changed()          # cached per-weight records (K20's absmax words) die with this update
        lib = _lib.load()
        self.steps += 1
        stream = torch.cuda.current_stream(ar.device).cuda_stream
        # fuse adjacent groups with the same hyper-parameters into one launch
        runs: List[list] = []
        for pg in self.param_groups:
            a, b = ar.segments[pg['segment']]
            hp = (float(pg['lr']), float(pg['betas'][0]), float(pg['betas'][1]), float(pg['eps']),
                  float(pg['weight_decay']))
            if runs and runs[-1][2] == hp and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b, hp])
        if self._k3_applied:
            # K3's backward already updated these two parameters (and left their gradient range untouched: zero): cut their
            # element ranges out of the launches
            self._k3_applied = False
            w, bp, ow, ob = self._k3
            for lo in sorted((ow, ob), reverse=True):
                hi = lo + _round_up(w.numel(), _ALIGN)
                cut = []
                for a, b, hp in runs:
                    if lo >= b or hi <= a:
                        cut.append([a, b, hp])
                    else:
                        if a < lo:
                            cut.append([a, lo, hp])
                        if hi < b:
                            cut.append([hi, b, hp])
                runs = cut
        sc = self.scaler
        if sc is not None:
            sc.check(ar.grad)
        for a, b, (lr, b1, b2, eps, wd) in runs:
            if b == a:
                continue
            sh = 0 if ar.shadow is None else ar.shadow.data_ptr() + 2 * a
            with ops.TIMER.span('k_adamw'):
                check(lib.mbv_adamw_step(ar.param.data_ptr() + 4 * a, ar.grad.data_ptr() + 4 * a,
                                         self.exp_avg.data_ptr() + 4 * a, self.exp_avg_sq.data_ptr() + 4 * a, sh,
                                         ar.shadow_flag, b - a, lr, b1, b2, eps, wd, self.steps,
                                         float(self.grad_scale), 1 if self.decoupled else 0,
                                         1 if self.zero_grad_in_step else 0,
                                         0 if sc is None else sc.scale.data_ptr(),
                                         0 if sc is None else sc.flag.data_ptr(),
                                         0 if sc is None else sc.applied_steps.data_ptr(), stream),
                      'mbv_adamw_step')
        if sc is not None:
            sc.update()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """Gradients live in the arena: clear in place (``step()`` already did when ``zero_grad=True``)."""
        if not self.zero_grad_in_step:
            self.arena.zero_grad()

    def state_dict(self):
        sd = super().state_dict()
        steps = self.steps if self.scaler is None else int(self.scaler.applied_steps.item())
        sd['flat_state'] = dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, steps=steps)
        if self.scaler is not None:
            sd['loss_scaler'] = self.scaler.state_dict()
        return sd

    def load_state_dict(self, sd):
        if self.scaler is not None and sd.get('loss_scaler') is not None:
            self.scaler.load_state_dict(sd['loss_scaler'])
        flat = sd.get('flat_state')
        if flat is not None:
            self.exp_avg.copy_(flat['exp_avg'])
            self.exp_avg_sq.copy_(flat['exp_avg_sq'])
            self.steps = int(flat['steps'])
            if self.scaler is not None:
                self.scaler.applied_steps.fill_(self.steps)
        elif sd.get('state'):
            # a torch.optim.Adam / AdamW state_dict (the reference's checkpoints, or a run saved without the arena):
            # its parameters are numbered in module.parameters() order = the order of the arena layout
            state = sd['state']
            if len(state) != len(self.arena.layout):
                raise ValueError(f'optimizer state has {len(state)} parameters, the arena {len(self.arena.layout)}: '
                                 'cannot map a per-parameter Adam state onto the arena')
            steps = 0
            with torch.no_grad():
                for idx, (p, o) in enumerate(self.arena.layout):
                    st = state[idx] if idx in state else state[str(idx)]
                    if tuple(st['exp_avg'].shape) != tuple(p.shape):
                        raise ValueError(f'optimizer state {idx} has shape {tuple(st["exp_avg"].shape)}, '
                                         f'parameter {tuple(p.shape)}')
                    n = p.numel()
                    self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
                    self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                    steps = max(steps, int(st['step']))
            self.steps = steps
            if self.scaler is not None:
                self.scaler.applied_steps.fill_(steps)
        for pg, saved in zip(self.param_groups, sd.get('param_groups', [])):
            for k in ('lr', 'betas', 'eps', 'weight_decay'):
                if k in saved:
                    pg[k] = saved[k]
