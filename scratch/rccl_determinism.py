"""Is the tiny model's graph step bit-reproducible from process to process, with and without the world-size-1 nccl reducer?
(tests/test_ddp_rccl_gpu.py's children; usage: python scratch/rccl_determinism.py [bf16|fp32])"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_ddp_rccl_gpu import _run
def main():
  dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
  with tempfile.TemporaryDirectory() as d:
      runs = {}
      for name, mode in (('none_a', 'none'), ('none_b', 'none'), ('f32_a', 'f32'), ('f32_b', 'f32')):
          p = pathlib.Path(d) / name
          p.mkdir()
          runs[name] = _run(p, mode, dt)
          print(name, runs[name]['losses'], flush=True)
      for a, b in (('none_a', 'none_b'), ('f32_a', 'f32_b'), ('none_a', 'f32_a')):
          print(a, b, 'max |d param|', float((runs[a]['params'] - runs[b]['params']).abs().max()))

if __name__ == '__main__':
  main()
