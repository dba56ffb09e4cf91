#!/usr/bin/env python3
"""Interleaved in-process A/B of the conv body alone (engine.conv_body on the bench images, no
head, nothing else on the device) over engine attributes and per-call environment knobs.

    python tools/ab_convbody.py --cases "" conv_streams=0 NAWS_CONV_RING=0
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic  # noqa: E402
from naws_hip import lib as L  # noqa: E402
from naws_hip.engine import WsddnEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', nargs='+', default=[''],
                    help='comma-separated name=value lists; UPPER-case names are environment knobs')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--images', type=int, default=2)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    eng = WsddnEngine(21, dev, gpu_num=a.images, seed=11)
    eng.set_conv_blobs(synthetic.init_blobs(20, seed=11))
    mb = synthetic.make_minibatch(synthetic.make_roidb(a.images, 16, 20, 600, 1000, seed=11), 20)
    data = torch.from_numpy(mb['data']).to(dev)
    base = {}
    cases = []
    for c in a.cases:
        kv = dict(x.split('=') for x in c.split(',') if x)
        cases.append(kv)
        for k in kv:
            if hasattr(eng, k):            # (engine attributes may be upper-case too: DIRECT_MIN_TILES)
                base.setdefault(k, getattr(eng, k))

    def apply(kv):
        for k, v in base.items():
            setattr(eng, k, v)
        for k in [k for c in cases for k in c if k.isupper() and not hasattr(eng, k)]:
            L.set_variant(L._ENV_KNOBS[k], {'NAWS_CONV_RING': 11, 'NAWS_ROI_NW': 42}.get(k, 0))
        for k, v in kv.items():
            if k.isupper() and not hasattr(eng, k):
                L.set_variant(L._ENV_KNOBS[k], int(v))
            else:
                setattr(eng, k, type(base[k])(int(v)))

    times = [[] for _ in cases]
    ref = None
    for r in range(a.rounds + 1):
        for i, kv in enumerate(cases):
            apply(kv)
            y = eng.conv_body(data)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                y = eng.conv_body(data)
            e.record()
            torch.cuda.synchronize()
            if r > 0:
                times[i].append(s.elapsed_time(e) / a.iters)
            if r == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    d = float((y - ref).abs().max() / ref.abs().max())
                    print('case %r: max |diff| / max |ref| vs the first case %.2e' % (a.cases[i], d))
    for i, c in enumerate(a.cases):
        ts = sorted(times[i])
        print('%-40s median %.3f ms (min %.3f max %.3f)' % (c or '(default)', ts[len(ts) // 2], ts[0], ts[-1]))


if __name__ == '__main__':
    main()
