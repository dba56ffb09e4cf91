#!/usr/bin/env python3
"""Interleaved in-process timing of the RoIPoolF operand-plane kernels on the bench shape
(2 images 74x124x512, 2 x 2000 proposals): direct vs hierarchical, waves per (roi, slice)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic  # noqa: E402
from naws_hip import lib as L, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    mb = synthetic.make_minibatch(synthetic.make_roidb(2, 2000, 20, 600, 1000, seed=11), 20)
    rois = torch.from_numpy(mb['rois']).to(dev)
    boost = torch.from_numpy(mb['obn_scores'].reshape(-1)).to(dev)
    x = torch.randn((2, 74, 124, 512), device=dev).relu_()
    amax = ops.amax_word(x).repeat(2)
    variants = [('direct', dict(hier=False), None)] + [
        ("hier nw*10+rg=%d" % nw, dict(hier=True), str(nw)) for nw in (42,)]
    times = {v[0]: [] for v in variants}
    ref = None
    for r in range(8):
        for name, kw, nw in variants:
            if nw is not None:
                L.set_variant('roi_nw', int(nw))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = ops.roi_pool_f_f16x2(x, rois, amax, 7, 7, 0.125, boost=boost, **kw)
            e.record()
            torch.cuda.synchronize()
            if r == 0:
                if ref is None:
                    ref = out.planes.clone()
                assert torch.equal(out.planes.view(torch.int16), ref.view(torch.int16)), name
            else:
                times[name].append(s.elapsed_time(e))
            del out
    for name, kw in (('fp32 direct', dict(hier=False)), ('fp32 hier', dict(hier=True))):
        ts = []
        for r in range(8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = ops.roi_pool_f(x, rois, 7, 7, 0.125, boost=boost, layout='NHWC', **kw)
            e.record()
            torch.cuda.synchronize()
            if r:
                ts.append(s.elapsed_time(e))
            del out
        times[name] = ts
    ws = torch.empty((2 * x.numel(),), device=dev)
    ts = []
    for r in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        # maps only: the hier entry with R = 0 rois is a no-op, so time the map build through a
        # 1-roi call
        out = ops.roi_pool_f(x, rois[:1], 7, 7, 0.125, boost=boost[:1], layout='NHWC', hier=True)
        e.record()
        torch.cuda.synchronize()
        if r:
            ts.append(s.elapsed_time(e))
    times['maps + 1 roi'] = ts
    for name, ts in times.items():
        ts = sorted(ts)
        print('%-20s median %.3f ms (min %.3f)' % (name, ts[len(ts) // 2], ts[0]))


if __name__ == '__main__':
    main()
