#!/usr/bin/env python3
"""dZ7 = gate(dL W8) on the bench shape (M = 4000, N = 4096, K = 40, batch 2): the fp32-MFMA form
against the register-resident FMA form it replaced (naws_set_variant('gemm', 8); only in the
`make AB=1` library: NAWS_LIB=.../libnaws_hip_ab.so, otherwise both rows time the MFMA form), with and without the
epilogue's |C| maxima, interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    rt, c2, n = 4000, 40, 4096
    dl = torch.randn((rt, 2 * c2), device=dev, generator=g) * 1e-3
    dlv = dl.view(rt, 2, c2).permute(1, 0, 2)
    w8 = torch.randn((2, c2, n), device=dev, generator=g) * 0.02
    h7 = torch.randn((rt, 2 * n), device=dev, generator=g)
    h7v = h7.view(rt, 2, n).permute(1, 0, 2)
    out = torch.empty_like(h7)
    outv = out.view(rt, 2, n).permute(1, 0, 2)
    res = {}
    ref = None
    for rnd in range(6):
        for var in (0, 8):
            for amax in (True, False):
                L.set_variant('gemm', var)
                kw = {}
                if amax:
                    sc_n, sc_t = ops.amax_scales(2, rt, dev), ops.amax_scales(2, n, dev)
                    kw = dict(rowmax=ops.amax_words(sc_n), colmax=ops.amax_words(sc_t))
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.gemm(dlv, w8, False, False, out=outv, epilogue=L.EPI_GATE_POS, aux=h7v, alpha=2.0, **kw)
                e.record()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                assert torch.allclose(out, ref, rtol=1e-4, atol=1e-7)
                if rnd:
                    res.setdefault((var, amax), []).append(s.elapsed_time(e))
    L.set_variant('gemm', 0)
    for (var, amax), ts in sorted(res.items()):
        ts = sorted(ts)
        print('%-22s maxima %-5s median %.1f us (min %.1f)' % ('fp32 MFMA 16x16x4' if var == 0 else 'FMA, B in registers',
                                                           amax, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
    print('bytes: %.0f MB (gate read + store) -> %.0f us at 4.5 TB/s' % (2 * out.numel() * 4 / 1e6, 2 * out.numel() * 4 / 4.5e6))


if __name__ == '__main__':
    main()
