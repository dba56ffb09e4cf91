#!/usr/bin/env python3
"""Soak test of the fused SGD + operand-plane kernel under real training dynamics: N steps at the
schedule's learning-rate scale on the bench workload; every `--check` steps the planes the SGD
kernel has been writing are compared with a from-scratch split of the current parameters, and
the overflow word (a row outgrew twice its previous maximum -> conditional re-split) is read.

    python tools/soak_update.py --steps 400 --lr 1e-4
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detectron.datasets import synthetic  # noqa: E402
from naws_hip import ops  # noqa: E402
from naws_hip.engine import WsddnEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--check', type=int, default=50)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--rois', type=int, default=500)
    ap.add_argument('--fused', type=int, default=1)
    ap.add_argument('--dump', default='')
    ap.add_argument('--train-step', type=int, default=1,
                    help='1: engine.train_step (one rank: fc6_w updated in its wgrad GEMM); 0: forward_backward + sgd_step')
    ap.add_argument('--mfma-dtype', default='fp16x2')
    ap.add_argument('--trace', type=int, default=0, help='print the loss every N steps')
    ap.add_argument('--exchange', type=int, default=0,
                    help='N > 1: the N-rank schedule with reducer.EmulatedExchange in the all-reduce\'s '
                         'place (gradients untouched): the deferred update runs piece by piece '
                         '(NAWS.PIPELINE_UPDATE) unless --pipeline 0; with --train-step 1 its parts are '
                         'queued from inside backward, as the training loop does')
    ap.add_argument('--pipeline', type=int, default=1)
    ap.add_argument('--switch-route', action='store_true',
                    help='with --exchange: change the update route at every check through '
                         'engine.set_update_route (piece by piece -> one launch -> piece by piece ...): '
                         'the digests must equal a run that stays on one route')
    ap.add_argument('--digest', action='store_true',
                    help='print a digest of the parameters / momentum / fc6_w planes at every check '
                         '(two runs that must be bit-identical print the same lines)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    c, B = 20, 2
    eng = WsddnEngine(c + 1, dev, gpu_num=B, seed=11, mfma_dtype=a.mfma_dtype)
    eng.fused_planes = bool(a.fused)
    if a.exchange > 1:
        from naws_hip.reducer import EmulatedExchange
        eng.reducer = EmulatedExchange(dev, a.exchange)
        eng.allreduce_chunks = 4 if a.exchange == 2 else 2
        eng.pipeline_update = bool(a.pipeline)
        print('exchange emulated for %d ranks; update piece by piece: %s' % (a.exchange, eng._pipelined()))
    blobs = synthetic.init_blobs(c, seed=11)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    eng.set_lr(a.lr)
    batches = []
    for s in range(4):
        mb = synthetic.make_minibatch(synthetic.make_roidb(B, a.rois, c, 320, 480, seed=11 + s), c)
        seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)).tolist()
        batches.append(({k: torch.from_numpy(v).to(dev) for k, v in mb.items()}, seg))
    tags = 0
    for it in range(a.steps):
        t, seg = batches[it % len(batches)]
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg,
                                   _fuse_update=bool(a.train_step))
        if a.trace and not bool(torch.isfinite(out['loss_cls'].sum() + out['loss_cls_noise'].sum())):
            for k, v in sorted(out.items()):
                v = v.float()
                bad = ~torch.isfinite(v)
                print('   %-20s shape %-16s non-finite %d  finite range [%.4g, %.4g]' % (
                    k, tuple(v.shape), int(bad.sum()), float(v[~bad].min()) if (~bad).any() else 0,
                    float(v[~bad].max()) if (~bad).any() else 0))
            if a.dump:
                torch.save({k: v.cpu() for k, v in out.items()} | {'labels_oh': t['labels_oh'].cpu(), 'rois': t['rois'].cpu(), 'seg': seg}, a.dump)
            return
        eng.sgd_step()
        if a.trace and (it + 1) % a.trace == 0:
            eng.flush()
            stat = ' '.join('%s %.3g/%.3g' % (n.replace('_[noisy]_', 'n_').replace('noisy_', 'n_'),
                                              float(eng.arena.view(eng.params, n).abs().max()),
                                              float(eng.arena.view(eng.grads, n).abs().max()))
                            for n, _ in eng.arena.specs if n.endswith('_w'))
            print('  %d: cls %.5f noise %.5f | max|w|/max|g| %s' % (
                it + 1, float(out['loss_cls'].sum()), float(out['loss_cls_noise'].sum()), stat), flush=True)
        if (it + 1) % a.check == 0:
            eng.flush()
            torch.cuda.synchronize()
            loss = float(out['loss_cls'].sum() + out['loss_cls_noise'].sum())
            ovf = int(eng._wovf.item()) if getattr(eng, '_wovf', None) is not None else 0
            tags += int(ovf == eng.sgd_iter_count)
            w6, w7 = eng._weight_views()
            worst = 0.0
            for key, w in ((('w6', w6), ('w7', w7)) if a.fused and a.mfma_dtype == 'fp16x2' else ()):
                got = eng._wplanes[key]
                p = got.planes.double()
                d = p[0] + p[1]
                d = d.unsqueeze(0) if d.dim() == 3 else d
                wd = w.reshape(-1, w.shape[-1]).double()
                dense = d.permute(0, 2, 1, 3).reshape(wd.shape) * got.inv_scale.reshape(-1).double()[:, None]
                rowmax = wd.abs().amax(dim=1, keepdim=True)
                err = ((dense - wd).abs() / torch.maximum(wd.abs() * 2.0 ** -22, rowmax * 2.0 ** -36)).max()
                worst = max(worst, float(err))
                fresh = ops.split_f16x2(w)
                ratio = got.inv_scale / fresh.inv_scale
                assert bool(((ratio == 1) | (ratio == 2)).all()), (it, key)
            if a.fused and a.mfma_dtype in ('bf16', 'fp32x3'):
                # no scales: the planes must BE the rounded / exactly split parameters
                cv = ops.to_bf16_slab if a.mfma_dtype == 'bf16' else ops.split_bf16x3
                for key, w, kw in (('w6', w6, {}), ('w7', w7, {}), ('w7t', w7, dict(transpose=True))):
                    assert torch.equal(eng._wplanes[key].view(torch.int16),
                                       cv(w, **kw).view(torch.int16)), (it, key)
            print('step %d: loss %.5f  max|w6| %.3f  last overflow tag %d (iter %d)  plane error / bound %.3f'
                  % (it + 1, loss, float(w6.abs().max()), ovf, eng.sgd_iter_count, worst), flush=True)
            if a.digest:
                import hashlib
                dg = [hashlib.blake2b(x.contiguous().view(torch.uint8).cpu().numpy(), digest_size=8).hexdigest()
                      for x in (eng.params, eng.momentum_buf, eng._wplanes['w6'].planes.view(torch.int16)
                                if a.mfma_dtype == 'fp16x2' and eng._wplanes else eng.params[:4])]
                print('   digest params %s momentum %s planes %s' % tuple(dg), flush=True)
            assert np.isfinite(loss) and worst <= 1.0
            if a.switch_route and a.exchange > 1:
                now = eng._pipelined()
                eng.set_update_route(pipeline_update=False if now else None)
                print('   route switched: update piece by piece %s -> %s' % (now, eng._pipelined()), flush=True)
    print('soak ok: %d steps, overflow re-splits seen at %d checkpoints' % (a.steps, tags))


if __name__ == '__main__':
    main()
