#!/usr/bin/env python
"""Static instruction mix of the generated gfx950 code, per kernel (runs here, no GPU needed).

    python tools/isa_count.py <file.hip> [substring of the mangled kernel name] [--ops] [-D...]

For every kernel of the file (or those whose mangled name contains the substring) this prints the number of vector
instructions by class (floating point / integer / compare-select / moves), scalar instructions, waits, LDS and
vector-memory operations, branches, full IEEE divisions (`v_div_fixup`), the register counts and spills from the
kernel's metadata, and with --ops the 40 most frequent opcodes.  A kernel that is bound by instruction issue runs as
fast as this count is small: round 5 found the 2-D row kernel's arg-max bookkeeping (280 compare-select, 653 scalar
instructions and 106 branches per thread and row) with it, and a run-time layout function that had put an integer
division and a branch in front of every load (profiles/r05_session3_notes.txt).  The count is static: both arms of a
branch are counted, so read it next to the source (and next to SQ_INSTS_VALU / SQ_WAVES of tools/pmc_cc2d.sh).
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'xcorr_amd', 'csrc')
FLAGS = ['-std=c++17', '-O3', '-fno-slp-vectorize', '-fPIC', '--offload-arch=gfx950', '-S', '--cuda-device-only']


def classify(op):
    if op.startswith('v_'):
        if op.startswith(('v_cmp', 'v_cndmask')):
            return 'v_cmpsel'
        if op.startswith(('v_mov', 'v_accvgpr')):
            return 'v_mov'
        if op.startswith('v_mfma') or op.startswith('v_smfma'):
            return 'mfma'
        if re.search(r'_(f16|f32|f64|bf16)', op) and not op.startswith('v_cvt'):
            return 'v_fp'
        return 'v_int'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop')):
            return 's_wait'
        if op.startswith('s_cbranch') or op == 's_branch':
            return 'branch'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('scratch_'):
        return 'scratch'
    if op.startswith(('global_', 'buffer_', 'flat_')):
        return 'vmem'
    return 'other'


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    defs = [a for a in sys.argv[1:] if a.startswith('-D')]
    show_ops = '--ops' in sys.argv
    if not args:
        raise SystemExit(__doc__)
    path, pat = args[0], (args[1] if len(args) > 1 else '')
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, *defs, '-o', tmp.name, os.path.abspath(path)], check=True, cwd=CSRC,
                       stderr=subprocess.DEVNULL)
        txt = open(tmp.name).read()
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
        meta[m.group(1)] = dict(re.findall(r'\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)', m.group(2)))
    for m in re.finditer(r'^(_Z\w+):.*?s_endpgm', txt, re.S | re.M):
        name = m.group(1)
        if pat not in name:
            continue
        c, ops = collections.Counter(), collections.Counter()
        for line in m.group(0).split('\n'):
            mm = re.match(r'^\s+([a-z][a-z0-9_]+)', line)
            if not mm:
                continue
            ops[mm.group(1)] += 1
            c[classify(mm.group(1))] += 1
        valu = c['v_fp'] + c['v_int'] + c['v_cmpsel'] + c['v_mov']
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        md = meta.get(name, {})
        print(dem[:150])
        print(f"   vector {valu} (fp {c['v_fp']}, int {c['v_int']}, compare-select {c['v_cmpsel']}, mov {c['v_mov']})  mfma {c['mfma']}  "
              f"scalar {c['salu']}  branches {c['branch']}  waits {c['s_wait']}  lds {c['lds']}  vmem {c['vmem']}  scratch {c['scratch']}  "
              f"divisions {ops['v_div_fixup_f32'] + ops['v_div_fixup_f64']}")
        print(f"   vgpr {md.get('vgpr_count', '?')} (spilled {md.get('vgpr_spill_count', '?')})  sgpr {md.get('sgpr_count', '?')}  "
              f"scratch bytes {md.get('private_segment_fixed_size', '?')}  static LDS {md.get('group_segment_fixed_size', '?')}")
        if show_ops:
            print('   ' + ', '.join(f'{k} {v}' for k, v in ops.most_common(40)))


if __name__ == '__main__':
    main()
