import torch
from mask_bev_amd import ops
from mask_bev_amd.arena import ParameterArena
from mask_bev_amd.layers import LayerNorm, Linear
torch.manual_seed(0)
dev='cuda'
lin=Linear(48,48).to(dev); ln=LayerNorm(48).to(dev)
import torch.nn as nn
mod=nn.ModuleList([lin,ln])
ref_lin=nn.Linear(48,48).to(dev); ref_ln=nn.LayerNorm(48).to(dev)
ref_lin.load_state_dict(lin.state_dict()); ref_ln.load_state_dict(ln.state_dict())
arena=ParameterArena([('all',mod)],shadow_dtype=None)
x=torch.randn(4,10,48,device=dev); h=torch.randn(4,10,48,device=dev)
d=ops.bias_grad_deferrable(lin.bias,48); print('deferrable',d)
y=ln(x, lin(h, skip_bias_grad=d), residual_bias=lin.bias if d else None)
y.square().sum().backward()
yr=ref_ln(x+ref_lin(h)); yr.square().sum().backward()
print('bias grad mine', lin.bias.grad[:5], 'ref', ref_lin.bias.grad[:5])
print('w grad diff', (lin.weight.grad-ref_lin.weight.grad).abs().max().item(), 'ln w', (ln.weight.grad-ref_ln.weight.grad).abs().max().item())
