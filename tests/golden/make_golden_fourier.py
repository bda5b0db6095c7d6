"""Golden vectors for A3 (per-point learnable Fourier encoding): the reference's own
``LearnableFourierPositionalEncoding`` (mask_bev/models/positional_encoding/learnable_fourier_positional_encoding.py,
importable in the build container: it needs only torch / numpy) run UNMODIFIED on seeded inputs with the two
group settings the encoder accepts (G = 1, M = 4 and G = 2, M = 2; mask_bev_encoders.py:51-58 builds it with
F_dim = 32, H_dim = 32, D = 128, gamma = 1).  Output: tests/golden/fourier.npz (inputs, parameters, outputs).
Runs only where /root/reference exists; the test-suite reads the committed file."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from mask_bev.models.positional_encoding.learnable_fourier_positional_encoding import (  # noqa: E402
    LearnableFourierPositionalEncoding)

out = {}
for g in (1, 2):
    torch.manual_seed(10 + g)
    enc = LearnableFourierPositionalEncoding(G=g, M=4 // g, F_dim=32, H_dim=32, D=128, gamma=1.0)
    x = torch.randn(37, g, 4 // g) * 3.0
    x[5] = 0.0                                     # a zero-padded slot of the dense voxel tensor
    with torch.no_grad():
        y = enc(x)
    out[f'g{g}_x'] = x.numpy()
    out[f'g{g}_y'] = y.numpy()
    for k, v in enc.state_dict().items():
        out[f'g{g}_{k}'] = v.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fourier.npz'), **out)
print({k: v.shape for k, v in out.items()})
