"""Torch-facing wrappers of the C-ABI kernels (``include/maskbev_hip.h``) — the ONE name the rest of the package uses.

PyTorch is plumbing here: it owns device memory, the stream and the autograd graph; every wrapper enqueues hand-written
gfx950 kernels on ``torch.cuda.current_stream()`` through ctypes.  There is no CPU fallback — tensors must live on a ROCm
device.  The wrappers live in one module per kernel family; this facade re-exports them (callers write ``ops.linear``,
tests patch ``ops.mask_logits`` / ``ops.hungarian`` here):

    ops_core       pointers, stream, dtype flags, the HIP-event timer, workspaces
    ops_records    fp32 mode: absmax records — pools, hints, static / weight / LayerNorm-bound registries, amax_verify
    ops_gemm       K17 / K20 GEMMs, Linear + FFN, the deferred / grouped parameter-gradient queues
    ops_encoder    K1 voxelise, K2 PillarFeatureNet, K3 scatter + LayerNorm
    ops_attention  K4 window attention, K6 decoder attention (+ shared K / V), K7 mask logits
    ops_msda       K5 / K16 multi-scale deformable attention
    ops_norm       K12 add + LayerNorm, bias + activation, K18 GroupNorm, patch merging
    ops_loss       K8 point sampling, K9 Hungarian, K10 importance sampling, K13 loss rows / costs
"""
from .ops_core import *            # noqa: F401,F403
from .ops_records import *         # noqa: F401,F403
from .ops_gemm import *            # noqa: F401,F403
from .ops_encoder import *         # noqa: F401,F403
from .ops_attention import *       # noqa: F401,F403
from .ops_msda import *            # noqa: F401,F403
from .ops_norm import *            # noqa: F401,F403
from .ops_loss import *            # noqa: F401,F403
