"""bench.py keeps the driver's contract: one JSON line with the agreed fields, the roofline object of the dominant
kernel and (when asked) the CPU baseline.  A 2-block encoder and 2 tiny steps keep this to seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--depth', '2', '--chunk', '4',
           '--no-both-modes', '--no-similarity', '--cpu-samples', '1']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['unit'] == 'point-clouds/s' and d['value'] > 0 and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert 'traffic' in r
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
