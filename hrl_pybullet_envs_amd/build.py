"""Build the gfx950 HIP library in-tree: hrl_pybullet_envs_amd/libhrl_envs_hip.so.

hipcc cross-compiles for gfx950 without a GPU.  Usage: python -m hrl_pybullet_envs_amd.build [--force]
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libhrl_envs_hip.so')
SOURCES = ['hrl_hip.hip', 'step_core.h', 'host_cfg.h']
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O2', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC', '-shared']


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/step_core.h + csrc/hrl_hip.hip) and the compiler flags: ties a committed counter summary
    (profiles/pmc_summary.json, tools/summarize_profile.py) to the code it was collected from (bench.py: roofline.pmc_stale)."""
    import hashlib
    h = hashlib.sha256()
    for name in ('step_core.h', 'hrl_hip.hip'):
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    h.update(' '.join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(PKG, '..', 'include', 'hrl_envs.h'), os.path.abspath(__file__)]  # this file holds the flags
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + HIPCC_FLAGS + ['-o', LIB, os.path.join(CSRC, 'hrl_hip.hip')]
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
