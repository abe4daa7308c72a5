#!/usr/bin/env python
"""Static check of the generated gfx950 code for loads the compiler serialised (runs here, no GPU needed).

Two patterns cost whole kernels their memory-level parallelism in this code base:
  * a load inside a bounds-check branch: every such load gets its own `s_waitcnt vmcnt(0)` before the branch joins
    (k2d_cols<inv>, round 4: 32 dependent round trips per thread);
  * a rolled loop `for (i = tid; i < n; i += NT) use(row[i])`: one load, one wait per iteration.
For every kernel of csrc/*.hip this prints the number of vector loads, of `s_waitcnt vmcnt(0)`, and the loops whose
body holds at most four loads and a full wait.  A kernel with about as many full waits as loads deserves a look at
its source; grid-stride loops of element-wise kernels are listed too and are harmless (one trip per thread).

    python tools/isa_scan.py [file.hip ...]        (default: every .hip of xcorr_amd/csrc except the instantiation stubs)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'xcorr_amd', 'csrc')
FLAGS = ['-std=c++17', '-O3', '-fno-slp-vectorize', '-fPIC', '--offload-arch=gfx950', '-S', '--cuda-device-only']


def demangle(name):
    try:
        return subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


def scan(path, extra=()):
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, *extra, '-o', tmp.name, path], check=True, cwd=CSRC,
                       stderr=subprocess.DEVNULL)
        txt = open(tmp.name).read()
    out = []
    for m in re.finditer(r'^(_Z\w+):.*?s_endpgm', txt, re.S | re.M):
        body = m.group(0).split('\n')
        labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
        loads = sum(1 for l in body if re.search(r'\b(global|buffer|flat)_load', l))
        full = sum(1 for l in body if 's_waitcnt vmcnt(0)' in l)
        loops = []
        for i, l in enumerate(body):
            mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if mm and labels.get(mm.group(1), i) < i:
                lb = body[labels[mm.group(1)]:i]
                nl = sum(1 for x in lb if re.search(r'\b(global|buffer|flat)_load', x))
                if 1 <= nl <= 4 and any('s_waitcnt vmcnt(0)' in x for x in lb):
                    loops.append((nl, len(lb)))
        out.append((demangle(m.group(1)), loads, full, loops))
    return out


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') and '_inst' not in f)
    for f in files:
        for name, loads, full, loops in scan(os.path.join(CSRC, f) if not os.path.isabs(f) else f):
            flag = '  <-- as many full waits as load pairs' if loads >= 16 and full >= loads // 3 else ''
            if loads >= 16 or loops:
                print(f'{os.path.basename(f):16s} loads {loads:4d}  vmcnt(0) {full:3d}  rolled loops (loads, lines) {loops[:4]}  '
                      f'{name[:110]}{flag}')


if __name__ == '__main__':
    main()
