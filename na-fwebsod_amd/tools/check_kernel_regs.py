#!/usr/bin/env python3
"""Register / spill counts of every kernel in csrc/*.hip at a git revision against the working
tree (no GPU needed: hipcc -S for gfx950 on both).  A kernel whose VGPR count or spill count moved
is printed - an edit to a shared template, argument struct or epilogue helper shows up here before it
shows up as an occupancy loss on the GPU (round 4: a lambda over the accumulators made the 256-row
gemm_x3_kernel forms spill ~300 registers; a tile loop took the SGD planes kernel from 96 to 130).

    python tools/check_kernel_regs.py [REV]        # default: HEAD
"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join('na-fwebsod_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas', '--offload-arch=gfx950',
         '-I.', '-S', '--cuda-device-only']


def meta(path):
    out = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', open(path).read(), re.S):
        v = re.search(r'\.vgpr_count:\s+(\d+)', m.group(2))
        sp = re.search(r'\.vgpr_spill_count:\s+(\d+)', m.group(2))
        if v:
            out[m.group(1)] = (int(v.group(1)), int(sp.group(1)) if sp else 0)
    return out


def compile_tree(d):
    def one(f):
        r = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', f[:-4] + '.s', f], cwd=d,
                           capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit('hipcc failed on %s' % f)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(one, sorted(os.path.basename(p) for p in glob.glob(os.path.join(d, '*.hip')))))


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else 'HEAD'
    with tempfile.TemporaryDirectory() as tmp:
        old, new = os.path.join(tmp, 'old'), os.path.join(tmp, 'new')
        os.makedirs(old), os.makedirs(new)
        names = subprocess.check_output(['git', 'ls-tree', '--name-only', rev, CSRC + '/'], cwd=ROOT,
                                        text=True).split()
        for n in names + ['include/naws.h']:
            if n.endswith(('.hip', '.h', '.inc')):
                open(os.path.join(old, os.path.basename(n)), 'wb').write(
                    subprocess.check_output(['git', 'show', '%s:%s' % (rev, n)], cwd=ROOT))
        for p in glob.glob(os.path.join(ROOT, CSRC, '*')) + [os.path.join(ROOT, 'include', 'naws.h')]:
            if p.endswith(('.hip', '.h', '.inc')):
                open(os.path.join(new, os.path.basename(p)), 'wb').write(open(p, 'rb').read())
        compile_tree(old), compile_tree(new)
        moved = 0
        for f in sorted(glob.glob(os.path.join(new, '*.s'))):
            b = os.path.basename(f)
            o = meta(os.path.join(old, b)) if os.path.exists(os.path.join(old, b)) else {}
            n = meta(f)
            same = sum(1 for k, v in n.items() if o.get(k) == v)
            for k, v in sorted(n.items()):
                if k in o and o[k] != v:
                    moved += 1
                    print('%s: %s\n    VGPRs / spilled: %s at %s -> %s now' % (b, k, o[k], rev, v))
            print('%-14s %3d kernels unchanged, %d not in %s (new or renamed)' % (
                b, same, sum(1 for k in n if k not in o), rev))
        return 1 if moved else 0


if __name__ == '__main__':
    sys.exit(main())
