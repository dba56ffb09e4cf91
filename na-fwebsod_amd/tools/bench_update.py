#!/usr/bin/env python3
"""Stand-alone time of the parameter update of the bench model (ACM SGD over the 957 MB head arena
+ the fp16x2 re-split of the updated fc6 / fc7 weights) with nothing else on the device: inside a
training step it runs on a side stream under the next conv body, where its events read longer."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic  # noqa: E402
from naws_hip.engine import WsddnEngine  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    eng = WsddnEngine(21, dev, gpu_num=2, seed=11)
    blobs = synthetic.init_blobs(20, seed=11)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    eng.set_lr(1e-5)
    # one training step first: the operand planes of the weights exist from then on
    mb = synthetic.make_minibatch(synthetic.make_roidb(2, 256, 20, 320, 480, seed=11), 20)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=2)).tolist()
    eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
    eng.sgd_step()
    eng.flush()
    eng.grads.normal_(0, 1e-3)
    nbytes = eng.params.numel() * 4

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    for knob, fp in (('fused_planes=1: ', True), ('fused_planes=0: ', False)):
        eng.fused_planes = fp
        eng.update_events = []
        t_all = timed(eng._apply_update)
        ev = eng.update_events[1:]
        t_sgd = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        eng.update_events = None
        print('%supdate %.3f ms = SGD %.3f ms (%.0f GB/s over 5 x %.0f MB) + re-split %.3f ms'
              % (knob, t_all, t_sgd, 5 * nbytes / t_sgd / 1e6, nbytes / 1e6, t_all - t_sgd))


if __name__ == '__main__':
    main()
