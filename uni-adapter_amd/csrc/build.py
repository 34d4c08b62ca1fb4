"""Build libuniadapter_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python uni-adapter_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(PKG, 'libuniadapter_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# -ffp-contract=off: fp32 element arithmetic follows the reference's separate torch ops
COMMON = ['-O3', f'--offload-arch={ARCH}', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall',
          '-Wno-unused-function']
PER_FILE = {}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(PKG), 'include', 'uniadapter.h'))
    hdrs.append(os.path.abspath(__file__))
    jobs = []
    for s in sources():
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, s[:-4] + '.o')
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + COMMON + PER_FILE.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
