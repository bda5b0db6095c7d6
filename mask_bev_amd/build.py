"""Builds ``libmaskbev_hip.so`` (gfx950) in-tree with hipcc.

``python -m mask_bev_amd.build`` or ``mask_bev_amd.build.build()``.  hipcc cross-compiles without a GPU,
so this also runs in the CPU-only build container; the resulting ``.so`` is git-ignored but travels to
the GPU box with the repository snapshot.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
BUILD_DIR = os.path.join(PKG_DIR, 'csrc', 'build')
LIB_PATH = os.path.join(PKG_DIR, 'libmaskbev_hip.so')
ARCH = 'gfx950'

# Per-file extra flags.  The voxeliser must keep `(p - min) / vs` as two IEEE f32 operations to stay
# bit-exact with the CPU reference, hence no contraction there (and never fast-math anywhere).
COMMON_FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-fno-fast-math', '-Wall',
                '-Wno-unused-function']
FILE_FLAGS = {
    'voxelize.hip': ['-ffp-contract=off'],
    'scatter_layernorm.hip': ['-ffp-contract=off'],
    # the LDS-DMA helper writes M0 inside its asm statement and says so in the clobber list (cdna_hip_programming.md §5.7)
    'gemm.hip': ['-Wno-inline-asm'],
}


def _hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libmaskbev_hip.so')
    return exe


def sources() -> List[str]:
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stamp(src: str, flags: List[str]) -> str:
    h = hashlib.sha1()
    h.update(' '.join(flags).encode())
    # `<name>_f16.hip` = the half instantiation of `<name>.hip` (it defines MBV_H16 and includes that file)
    twin = [src.replace('_f16.hip', '.hip')] if src.endswith('_f16.hip') else []
    for name in [src] + twin + sorted(f for f in os.listdir(CSRC) if f.endswith('.hpp')) + \
            ['../../include/maskbev_hip.h']:
        with open(os.path.join(CSRC, name), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile_one(src: str, verbose: bool) -> str:
    flags = COMMON_FLAGS + FILE_FLAGS.get(src.replace('_f16.hip', '.hip'), [])
    obj = os.path.join(BUILD_DIR, src.replace('.hip', '.o'))
    stamp_file = obj + '.stamp'
    stamp = _stamp(src, flags)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    LAST_BUILD['recompiled'].append(src)
    cmd = [_hipcc(), '-c', os.path.join(CSRC, src), '-o', obj] + flags
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, 'w') as fh:
        fh.write(stamp)
    return obj


# what the last build() call did: 'stamp hit' (every object's source + flag hash matched, nothing compiled),
# 'recompiled' (the listed translation units were compiled) — printed by __graft_entry__.build()
LAST_BUILD = {'mode': None, 'recompiled': [], 'linked': False}


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(BUILD_DIR, exist_ok=True)
    LAST_BUILD.update(mode=None, recompiled=[], linked=False)
    if force:
        for f in os.listdir(BUILD_DIR):
            os.remove(os.path.join(BUILD_DIR, f))
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, verbose), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        cmd = [_hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        LAST_BUILD['linked'] = True
    LAST_BUILD['mode'] = 'recompiled' if LAST_BUILD['recompiled'] else ('relinked' if LAST_BUILD['linked'] else 'stamp hit')
    return LIB_PATH


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
