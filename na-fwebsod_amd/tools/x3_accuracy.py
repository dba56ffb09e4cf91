#!/usr/bin/env python3
"""Measured accuracy of the split-operand fp32 GEMMs - fp16x2 (row-scaled 2-way f16 split, 3 MFMA
passes) and fp32x3 (exact 3-way bf16 split, 6 passes) - next to the fp32-MFMA GEMM and the bf16
GEMM, all against a float64 product of the same fp32 operands.
Prints a markdown table (kept as profiles/rNN_x3_accuracy.md)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    print('| operands | M x N x K | fp16x2 max / rms | fp32x3 max / rms | fp32 MFMA max / rms | bf16 max / rms |')
    print('|---|---|---|---|---|---|')
    cases = [('uniform(-1,1)', 512, 512, 4096, None), ('uniform(-1,1)', 256, 256, 25088, None),
             ('normal x row scales e^U(-14,14)', 384, 320, 2048, 14.0),
             ('ReLU-like (half zeros) x normal', 512, 512, 4096, 'relu'),
             ('ReLU-like x |normal| (same-sign sums)', 384, 320, 8192, 'pos'),
             ('ReLU-like x |normal| (same-sign sums)', 256, 256, 25088, 'pos')]
    bias_rows = []
    for name, m, n, k, kind in cases:
        a = rng.uniform(-1, 1, (m, k)) if kind is None else rng.standard_normal((m, k))
        b = rng.uniform(-1, 1, (n, k)) if kind is None else rng.standard_normal((n, k))
        if isinstance(kind, float):
            a *= np.exp(rng.uniform(-kind, kind, (m, 1)))
            b *= np.exp(rng.uniform(-kind, kind, (n, 1)))
        if kind in ('relu', 'pos'):
            a = np.maximum(a, 0)
        if kind == 'pos':
            a *= np.exp(rng.uniform(-2, 2, (m, k)))
            b = np.abs(b) * 0.01
        a, b = a.astype(np.float32), b.astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64).T
        # per-entry error relative to the entry's own scale |a|.|b| (fp32 dot-product bound)
        bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
        ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        outs = [ops.gemm_f32_f16x2_nt(ops.split_f16x2(ad), ops.split_f16x2(bd)),
                ops.gemm_f32x3_nt(ops.split_bf16x3(ad), ops.split_bf16x3(bd)),
                ops.gemm(ad, bd, False, True),
                ops.gemm_bf16_slab_nt(ops.to_bf16_slab(ad), ops.to_bf16_slab(bd))]
        cells = []
        for o in outs:
            e = np.abs(o.cpu().numpy().astype(np.float64) - ref) / bound
            cells.append('%.2e / %.2e' % (e.max(), np.sqrt(np.mean(e ** 2))))
        print('| %s | %d x %d x %d | %s | %s | %s | %s |' % (name, m, n, k, *cells))
        if kind == 'pos':
            bias_rows.append('| %d | ' % k + ' | '.join(
                '%+.2e' % np.mean((o.cpu().numpy().astype(np.float64) - ref) / ref) for o in outs[:3])
                + ' | %+.2e |' % np.mean(((ad @ bd.t()).cpu().numpy().astype(np.float64) - ref) / ref))
    print()
    print('Mean signed relative error on the same-sign cases (accumulation bias; the 16-bit MFMAs '
          'round their internal sums toward zero, whichever way the operands were split):')
    print()
    print('| K | fp16x2 | fp32x3 | fp32 MFMA | hipBLAS sgemm |')
    print('|---|---|---|---|---|')
    for r in bias_rows:
        print(r)
    print()
    print('Errors are |c - c64| / (|a|.|b|) per output entry (max and rms over the matrix); '
          'fp32 unit roundoff is 6.0e-08, bf16 3.9e-03.')


if __name__ == '__main__':
    main()
