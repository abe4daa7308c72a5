#!/usr/bin/env python
"""Static check of the generated gfx950 code for loads the compiler serialised (runs here, no GPU needed).

Two patterns cost whole kernels their memory-level parallelism in this code base:
  * a load inside a bounds-check branch: every such load gets its own `s_waitcnt vmcnt(0)` before the branch joins
    (k2d_cols<inv>, round 4: 32 dependent round trips per thread);
  * a rolled loop `for (i = tid; i < n; i += NT) use(row[i])`: one load, one wait per iteration.
Round 5 added three more shapes, all found in the pair kernels' generated code ("Round 5, second session"):
  * memory operations on a 64-bit VECTOR address (`global_load ... v[a:b], off`) where a scalar base + 32-bit per-lane
    offset would do: the optimiser reassociates base + constant + offset into (base + offset) + constant, and every load
    gets a `v_lshl_add_u64` and a register pair of its own (column `vaddr64`, against `saddr`);
  * a wave-uniform value fetched with a VECTOR load and moved to a scalar register (`global_load` ... `v_readfirstlane`
    of the loaded register): a round trip through the CU's vector-memory pipeline in front of whatever depends on it
    (column `rfl-loads`; the constant address space makes it an `s_load`);
  * loops around an LDS read and a full `s_waitcnt lgkmcnt(0)` (column `lds-loops`).
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
        vaddr64 = sum(1 for l in body if re.match(r'\s*(global|flat)_(load|store)\w*\s+.*\boff\b', l))
        saddr = sum(1 for l in body if re.match(r'\s*global_(load|store)', l)) - vaddr64
        # vector load whose destination register is read by v_readfirstlane before anything overwrites it
        loaded, rfl = {}, 0
        for i, l in enumerate(body):
            ls = l.strip()
            mm = re.match(r'(global|buffer|flat)_load_\w+\s+(?:v\[(\d+):(\d+)\]|v(\d+))', ls)
            if mm:
                for v in (range(int(mm.group(2)), int(mm.group(3)) + 1) if mm.group(2) else [int(mm.group(4))]):
                    loaded[v] = i
                continue
            mm = re.match(r'v_readfirstlane_b32\s+s\d+,\s*v(\d+)', ls)
            if mm and int(mm.group(1)) in loaded and i - loaded[int(mm.group(1))] < 80:
                rfl += 1
                continue
            mm = re.match(r'v_\w+\s+v(\d+)', ls)
            if mm:
                loaded.pop(int(mm.group(1)), None)
        lds_loops = 0
        for i, l in enumerate(body):
            mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
            tgt = mm and (mm.group(1) or mm.group(2))
            if tgt and labels.get(tgt, i) < i:
                lb = body[labels[tgt]:i]
                if any('ds_read' in x for x in lb) and any(re.search(r's_waitcnt.*lgkmcnt\(0\)', x) for x in lb) and len(lb) < 400:
                    lds_loops += 1
        out.append((demangle(m.group(1)), loads, full, loops, vaddr64, saddr, rfl, lds_loops))
    return out


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') and '_inst' not in f)
    for f in files:
        for name, loads, full, loops, vaddr64, saddr, rfl, lds_loops in scan(os.path.join(CSRC, f) if not os.path.isabs(f) else f):
            flag = '  <-- as many full waits as load pairs' if loads >= 16 and full >= loads // 3 else ''
            if vaddr64 >= 12 and vaddr64 > saddr:
                flag += '  <-- mostly 64-bit vector addresses'
            if rfl:
                flag += '  <-- uniform value through a vector load'
            if loads >= 16 or loops or rfl or lds_loops:
                print(f'{os.path.basename(f):16s} loads {loads:4d}  vmcnt(0) {full:3d}  vaddr64 {vaddr64:3d}  saddr {saddr:3d}  rfl-loads {rfl}  '
                      f'lds-loops {lds_loops}  rolled loops (loads, lines) {loops[:4]}  {name[:100]}{flag}')


if __name__ == '__main__':
    main()
