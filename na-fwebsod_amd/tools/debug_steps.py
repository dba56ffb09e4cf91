#!/usr/bin/env python3
"""Print per-iteration losses and blob magnitudes of the synthetic bench workload."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic
from naws_hip.engine import WsddnEngine

dev = torch.device('cuda:0')
c = 20
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = WsddnEngine(c + 1, dev, gpu_num=2, seed=11)
blobs = synthetic.init_blobs(c, seed=11)
eng.set_conv_blobs(blobs); eng.set_head_blobs(blobs)
mb = synthetic.make_minibatch(synthetic.make_roidb(2, R, c, 600, 1000, seed=11), c)
t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
eng.set_lr(1e-3)
for it in range(6):
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    print(it, 'loss', out['loss_cls'].tolist(), out['loss_cls_noise'].tolist())
    for k in ('cls_prob', 'class_weight', 'hatE_sum', 'hatE_sum_norm', 'd_logits', 'rois_pred'):
        v = out[k]
        print('   ', k, 'min %.3e max %.3e nan %d' % (float(v[~v.isnan()].min()), float(v[~v.isnan()].max()), int(v.isnan().sum())))
    print('    grads absmax', float(eng.grads.abs().max()), 'params absmax', float(eng.params.abs().max()))
    eng.sgd_step()
