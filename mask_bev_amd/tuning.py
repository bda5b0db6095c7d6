"""Library-GEMM solution selection for the MaskBEV step on MI355X.

The step's GEMMs (Swin / pixel-decoder / decoder Linears and their data / weight gradients) run on hipBLASLt through
torch.  Its default heuristic is not the fastest solution for a third of this model's shapes (few-thousand-row
token matrices against 192…3072-wide weights; weight gradients with K = tokens): ``tuned/gemm_gfx950.csv`` holds the
solution index that PyTorch's TunableOp measured fastest for each of the 108 GEMM signatures of the
``semantic_kitti_512`` step at 4 scans per GPU (38.9 → 38.1 ms per step).  Only the *selection* changes — every
entry is a stock hipBLASLt solution (or ``Default``).  Round 4 added the 100 signatures of the same step in fp32
(``bench.py --dtype fp32`` under the command below, starting from the committed file): the fp32 step is library-GEMM
bound (26.7 of 55 ms) and went from 72.6 to 75.7 scans/s with the measured selections (93 of the 100 are not the default);
the fp16 step's 55 signatures were added the same way (fp16 now equals bf16 within the noise).  Re-tuning the bf16 step
from the committed table hit a hipBLASLt candidate that faults on this ROCm (memory access fault, like the pinned
``tn_256_100_16384_B_4``): the bf16 entries are the earlier rounds'.

``use_tuned_gemms()`` switches TunableOp on in look-up-only mode.  Shapes that are not in the table, or a table
whose validator lines (PyTorch / ROCm / hipBLASLt versions, GPU architecture) do not match the running stack, fall
back to the library default, so the call is always safe.

Re-generating the table (one GPU, ≈ 1 minute):

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_ROCBLAS_ENABLED=0 \\
    PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=12 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 \\
    PYTORCH_TUNABLEOP_FILENAME=/tmp/gemm.csv python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
    cp /tmp/gemm0.csv mask_bev_amd/tuned/gemm_gfx950.csv

(one strided-batched bf16 signature, ``tn_256_100_16384_B_4``, is pinned to ``Default`` in the table: a hipBLASLt
candidate for it faults during tuning on ROCm 7.0 — start from the committed file so that it is skipped.)
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import switches

DEFAULT_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned', 'gemm_gfx950.csv')


def use_tuned_gemms(table: Optional[str] = None) -> bool:
    """Look GEMM solutions up in ``table`` (default: the committed gfx950 table).  Returns whether the table was
    accepted.  ``switches.tuned_gemms = False`` disables it; an explicit ``PYTORCH_TUNABLEOP_ENABLED`` in the environment is
    left alone (the user is driving TunableOp themselves)."""
    if not switches.get('tuned_gemms') or 'PYTORCH_TUNABLEOP_ENABLED' in os.environ:
        return False
    if not torch.cuda.is_available():
        return False
    table = table or DEFAULT_TABLE
    if not os.path.isfile(table):
        return False
    from torch.cuda import tunable
    tunable.enable(True)
    tunable.tuning_enable(False)                 # look-up only: never tune (or write files) inside a job
    tunable.record_untuned_enable(False)
    try:
        tunable.write_file_on_exit(False)
    except AttributeError:
        pass
    ok = bool(tunable.read_file(table))
    if not ok:
        tunable.enable(False)
    return ok
