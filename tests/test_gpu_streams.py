"""GPU: the two launch pipelines of the pair phase keep hardware queues of their own whatever other plans the
process created before (the pipeline streams belong to the context and are created with it).  With per-plan
streams a correlation plan created AFTER a time-shift plan and the any-length-transform plans ran both
pipelines on one queue and took 1.33 x the time.  tools/streams_check.py times the station-day step in a fresh
process with and without those plans created first; the two must stay within 15 %."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(order):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'streams_check.py'), order, '32'],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return float(re.search(r'ms_per_step ([0-9.]+)', out.stdout).group(1))


def test_pipelines_overlap_after_other_plans():
    fresh, late = _run('fresh'), _run('before')
    assert late < 1.15 * fresh, f'pair pipelines lost their overlap: {fresh:.1f} -> {late:.1f} ms per step'
