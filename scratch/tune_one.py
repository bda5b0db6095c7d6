"""gpurun helper (scratch/tune_cmd.sh): tune the GEMM signatures of one file in THIS process and write the selections.
python scratch/tune_one.py UNTUNED.csv OUT.csv [rotating_buffer_MB]   — a hipBLASLt candidate that faults takes only this
process down (the driver script then pins that signature to Default)."""
import sys
import torch
from torch.cuda import tunable

src, out = sys.argv[1], sys.argv[2]
rot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tunable.enable(True)
tunable.tuning_enable(True)
tunable.record_untuned_enable(False)
tunable.set_max_tuning_duration(25)
tunable.set_max_tuning_iterations(200)
if rot:
    tunable.set_rotating_buffer_size(rot)          # operands rotate through `rot` MB: candidates are timed cache-cold
tunable.set_filename(out)
tunable.tune_gemm_in_file(src)          # (this PyTorch appends every selection to `out` as it is found: no write call)
torch.cuda.synchronize()
print('tuned', len([r for r in tunable.get_results()]), 'signatures ->', out)
