#!/usr/bin/env python3
"""Interleaved in-process A/B of two builds of libnaws_hip.so on the hot GEMM / conv shapes
(perf deltas between separate runs or boxes are not comparable: DVFS and device spread).

    python tools/ab_gemm.py lib/ab/libA.so lib/ab/libB.so [--rounds 7]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L  # noqa: E402


def bind(path):
    l = C.CDLL(path)
    for name in ('naws_gemm_f32', 'naws_conv3x3_nhwc_fwd'):
        getattr(l, name).argtypes = L.PROTOTYPES[name]
        getattr(l, name).restype = C.c_int
    return l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--rows', type=int, default=4000)
    ap.add_argument('--env', nargs='*', default=[],
                    help='NAME=VALUE applied while lib i makes its FIRST call (one per lib, "-" = none): '
                         'the variant knob is latched per library on first use')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    libs = [bind(p) for p in a.libs]
    st = torch.cuda.current_stream().cuda_stream
    R = a.rows

    def rnd(*s):
        return torch.empty(s, device=dev).uniform_(-1, 1)

    cases = []
    for name, m, n, k, ta, tb, batch in [
            ('fc6 fwd NT', R, 8192, 25088, 0, 1, 1), ('fc7 fwd NT', R, 4096, 4096, 0, 1, 2),
            ('fc7 dgrad NN', R, 4096, 4096, 0, 0, 2), ('fc7 wgrad TN', 4096, 4096, R, 1, 0, 2),
            ('fc6 wgrad TN', 8192, 25088, R, 1, 0, 1)]:
        A = rnd(batch, *((k, m) if ta else (m, k)))
        B = rnd(batch, *((n, k) if tb else (k, n)))
        Cc = torch.empty((batch, m, n), device=dev)
        args = (ta, tb, m, n, k, A.data_ptr(), A.stride(1), B.data_ptr(), B.stride(1),
                Cc.data_ptr(), n, batch, A.stride(0), B.stride(0), Cc.stride(0), 0, None, 0, None,
                0, 1.0, 0.0, 0, 0, st)
        cases.append((name, 'naws_gemm_f32', args, 2.0 * m * n * k * batch, (A, B, Cc)))
    for cin, cout, h, w, dil in [(64, 64, 600, 1000, 1), (128, 128, 300, 500, 1),
                                 (256, 256, 150, 250, 1), (512, 512, 75, 125, 1),
                                 (512, 512, 74, 124, 2)]:
        x, wp, b = rnd(2, h, w, cin), rnd(cout, 3, 3, cin), rnd(cout)
        y = torch.empty((2, h, w, cout), device=dev)
        args = (x.data_ptr(), wp.data_ptr(), b.data_ptr(), 2, h, w, cin, cout, dil, 1,
                y.data_ptr(), st)
        cases.append(('conv %d->%d %dx%d d%d' % (cin, cout, h, w, dil), 'naws_conv3x3_nhwc_fwd',
                      args, 2.0 * 2 * h * w * cout * 9 * cin, (x, wp, b, y)))
    first = [True] * len(libs)
    for name, fn, args, flops, _keep in cases:
        times = [[] for _ in libs]
        for r in range(a.rounds + 1):
            for i, l in enumerate(libs):
                if first[i] and i < len(a.env) and a.env[i] != '-':
                    k, v = a.env[i].split('=')      # NAWS_GEMM_VARIANT=4 -> this build's knob
                    l.naws_set_variant.argtypes = [C.c_char_p, C.c_int]
                    assert l.naws_set_variant(L._ENV_KNOBS[k].encode(), int(v)) == 0
                first[i] = False
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = getattr(l, fn)(*args)
                e.record()
                torch.cuda.synchronize()
                assert rc == 0, rc
                if r > 0:
                    times[i].append(s.elapsed_time(e))
        med = [sorted(t)[len(t) // 2] for t in times]
        mn = [min(t) for t in times]
        print('%-26s ' % name + '  '.join('%s: med %.3f ms %.1f TF (min %.3f)' % (
            os.path.basename(p), m_, flops / m_ / 1e9, n_) for p, m_, n_ in zip(a.libs, med, mn)))


if __name__ == '__main__':
    main()
