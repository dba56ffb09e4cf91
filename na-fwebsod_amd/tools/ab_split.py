#!/usr/bin/env python3
"""Interleaved in-process A/B of the operand-plane split pass (`naws_split_f16x2_dual`): the
16-byte form (knob "split" = 2; the default 0 takes it for the transposed planes alone) against
the scalar form (1) on the three shapes of the
head's main stream (h6 and dZ7 -> both forms, dZ6 -> transposed only), planes compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    cases = [('h6 / dZ7: rows 4000 x 4096, both forms', 4000, 4096, True, True, False),
             ('dZ6: 4000 x 8192, transposed, rowmul', 4000, 8192, False, True, True),
             ('ragged 1999 x 4100, both', 1999, 4100, True, True, True)]
    for name, rows, cols, wn, wt, rm in cases:
        x = torch.randn((rows, cols), device=dev) * torch.rand((rows, 1), device=dev).exp()
        rowmul = (torch.rand((rows,), device=dev) + 0.5) if rm else None
        xs = x * rowmul[:, None] if rm else x
        res, times = {}, {2: [], 1: []}
        for rep in range(9):
            for knob in (2, 1):
                L.set_variant('split', knob)
                sn = st = None
                if wn:
                    sn = torch.zeros((2, rows), device=dev)
                    sn[0] = x.abs().amax(1)
                if wt:
                    st = torch.zeros((2, cols), device=dev)
                    st[0] = xs.abs().amax(0)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                pn, pt = ops.split_f16x2_dual(x, scales_n=sn, scales_t=st, rowmul=rowmul)
                e.record()
                torch.cuda.synchronize()
                if rep == 0:
                    res[knob] = [t.planes.view(torch.int16).clone() if t is not None else None for t in (pn, pt)] + \
                        [t.scales.clone() if t is not None else None for t in (pn, pt)]
                else:
                    times[knob].append(s.elapsed_time(e))
                del pn, pt
        for a, b in zip(res[2], res[1]):
            assert (a is None and b is None) or torch.equal(a, b), name
        L.set_variant('split', 0)
        byt = rows * cols * 4 * (1 + int(wn) + int(wt))
        for knob in (2, 1):
            t = sorted(times[knob])[len(times[knob]) // 2]
            print('%-44s %s  %.1f us  %.2f TB/s' % (name, {2: '16-byte', 1: 'scalar '}[knob], t * 1e3, byt / t / 1e9))


if __name__ == '__main__':
    main()
