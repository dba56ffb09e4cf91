#!/usr/bin/env python3
"""Interleaved in-process A/B of ONE engine attribute (or environment knob read per call) on the
bench workload: blocks of training steps alternate between the values, same process, same device,
same data (DVFS and device spread make numbers from separate runs incomparable:
cdna_hip_programming.md rule 24).

    python tools/ab_engine.py --attr fused_planes --values 1 0 [--rounds 6 --steps 10]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic  # noqa: E402
from naws_hip.engine import WsddnEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--attr', default='')
    ap.add_argument('--env', default='', help='environment knob read per call instead of an attribute')
    ap.add_argument('--values', nargs='+', required=True)
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--rois', type=int, default=2000)
    ap.add_argument('--mfma-dtype', default='fp16x2')
    ap.add_argument('--fix', nargs='*', default=[], help='attr=value held for the whole run')
    ap.add_argument('--stages', action='store_true', help='per-stage times (HIP events on the main stream)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    c, B = 20, 2
    eng = WsddnEngine(c + 1, dev, gpu_num=B, seed=11, mfma_dtype=a.mfma_dtype)
    blobs = synthetic.init_blobs(c, seed=11)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    del blobs
    mb = synthetic.make_minibatch(synthetic.make_roidb(B, a.rois, c, 600, 1000, seed=11), c)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    counts = np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)
    seg = [0] + np.cumsum(counts).tolist()
    eng.set_lr(1e-5)

    def conv(v):
        try:
            return int(v)
        except ValueError:
            return v

    for kv in a.fix:
        k, v = kv.split('=')
        if not hasattr(eng, k):
            raise SystemExit('no engine attribute %r' % k)
        setattr(eng, k, conv(v))

    host = []

    def run(n):
        h0 = time.perf_counter()
        for _ in range(n):
            eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
        eng.flush()
        host.append((time.perf_counter() - h0) / n * 1e3)     # enqueue time, before the GPU is done
        torch.cuda.synchronize()

    vals = [conv(v) for v in a.values]
    stages = {}
    times = {v: [] for v in vals}
    run(5)
    for _ in range(a.rounds):
        for v in vals:
            if a.env:
                from naws_hip import lib as _L
                _L.set_variant(_L._ENV_KNOBS.get(a.env, a.env), int(v))
            else:
                setattr(eng, a.attr, v)
            run(3)
            if a.stages:
                eng.phase_events = []
            t0 = time.perf_counter()
            run(a.steps)
            times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
            if a.stages:
                ev, eng.phase_events = eng.phase_events, None
                for (_n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                    if n1 != 'start':
                        stages.setdefault(v, {}).setdefault(n1, []).append(e0.elapsed_time(e1))
    print('host enqueue time per step: median %.3f ms' % sorted(host)[len(host) // 2])
    for v in vals:
        ts = sorted(times[v])
        print('%s=%r: median %.3f ms/step (min %.3f, max %.3f)' % (a.env or a.attr, v, ts[len(ts) // 2], ts[0], ts[-1]))
        if v in stages:
            print('    ' + '  '.join('%s %.3f' % (n, float(np.median(x))) for n, x in stages[v].items()))


if __name__ == '__main__':
    main()
