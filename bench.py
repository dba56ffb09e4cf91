#!/usr/bin/env python3
"""Headline benchmark: images/sec of the NA-fWebSOD hot path (VGG16-C5 conv body ->
RoIPoolF(+boost) -> two 2-fc branches -> WSDDN dual softmax -> entropy-gated weighted
CE -> backward -> gradient all-reduce -> ACM momentum SGD), fp32, on synthetic
600x1000 images with 2000 proposals each (BASELINE.json configs[1]; configs[2] at
--gpus 8: batch = 2 images per GPU).

A "step" = one full training iteration of every rank on its own 2 images.  Inputs are
resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, dense v_mfma_f32_32x32x16_bf16 / _f16
HBM_ACHIEVABLE_GBPS = 6300.0    # same guide: ~6.3 TB/s achievable of the 8 TB/s HBM3E spec
LINK_GBPS = 153.0               # same guide: one xGMI link, per direction; 7 links per GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150,
                    help='150 x ~14 ms: a timed region of >= 2 s (the clock of the power-limited '
                         'GEMMs settles over hundreds of ms)')
    ap.add_argument('--warmup', type=int, default=20,
                    help='SURVEY.md 8(d): discard the first 20 iterations')
    ap.add_argument('--images-per-gpu', type=int, default=2)
    ap.add_argument('--rois', type=int, default=2000)
    ap.add_argument('--height', type=int, default=600)
    ap.add_argument('--width', type=int, default=1000)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--lr', type=float, default=1e-5,
                    help='synthetic Kaiming weights collapse the MIL softmax within a few '
                         'iterations at the schedule lr 1e-3 (entropy gate -> 0/0, as in the '
                         'reference); the SGD work is lr-independent')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL and run the all-reduce schedule even with one rank '
                         '(exercises the N>1 code path on a 1-GPU box)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='DIAGNOSTIC: every rank uses cuda:0 and the gradients travel over gloo '
                         '(RCCL refuses two ranks on one device): runs the real N > 1 schedule - '
                         'chunked fc6_w exchange, deferred update behind it - on a 1-GPU box.  The '
                         'line says shared_gpu: true; its value is NOT a throughput claim')
    ap.add_argument('--emulate-exchange', default='',
                    help='N[,CUS[,GB/s]]: one rank only - run the N-rank schedule (gradient written, '
                         'chunked fc6_w messages, deferred update) with naws_emulate_exchange '
                         'standing in for the RCCL all-reduce: CUS compute units (default 32) '
                         'move 2 (N-1)/N of every message through HBM at the links\' pace '
                         '(default 0.6 x 153 GB/s x min(N-1, 7)).  The line is labelled a '
                         'projection.  The default one-GPU run appends the N = 8 projection')
    ap.add_argument('--sharded-update', action='store_true',
                    help='N > 1: NAWS.SHARDED_UPDATE - fc6_w gradient rows reduced to one owner each, '
                         'owner-only update, updated rows all-gathered (engine._apply_update_sharded)')
    ap.add_argument('--no-pipeline-update', action='store_true',
                    help='N > 1: wait for the whole gradient exchange and update in one launch '
                         '(NAWS.PIPELINE_UPDATE False) instead of piece by piece with fc6 forward '
                         'starting behind each piece')
    ap.add_argument('--drop-sharded-update', action='store_true',
                    help='cancels --sharded-update (the supervisor\'s fallback attempts append flags '
                         'to the original command line)')
    ap.add_argument('--no-supervisor', action='store_true',
                    help='N > 1: run the rank in THIS process instead of as a watched child of a '
                         'per-rank supervisor (naws_hip/supervise.py: per-phase watchdog, fresh '
                         'workers on the unpipelined route after a stall, a death or a rank-digest '
                         'mismatch)')
    ap.add_argument('--no-route-ab', action='store_true',
                    help='N > 1: skip the in-run A/B of the update routes after the timed region '
                         '(value_unpipelined / value_pipelined, value_sharded)')
    ap.add_argument('--no-projection', action='store_true',
                    help='skip the N-rank contention projections (profile runs: their proxy kernel '
                         'and chunked wgrad launches would otherwise sit in the kernel tables)')
    ap.add_argument('--no-fused-update', action='store_true',
                    help='one rank only: write fc6_w\'s gradient and update it in the deferred SGD '
                         'kernel (the route every rank takes when there is a gradient exchange) '
                         'instead of in the wgrad GEMM\'s epilogue')
    ap.add_argument('--self-launch', action='store_true',
                    help='start the ranks as children through torch.distributed.run even for '
                         '--gpus 1 (the path a bare `python bench.py --gpus N`, N > 1, always takes)')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU work: the ranks rendezvous over gloo and run the barrier / '
                         'max-over-ranks / one-JSON-line skeleton with a 1 ms sleep as the step '
                         '(checks the launcher and the multi-rank plumbing on a CPU box; the line '
                         'is labelled dry-run and carries no throughput claim)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=10,
                    help='timed iterations of the CPU restatement (BASELINE.md section 3: 10)')
    ap.add_argument('--cpu-warmup', type=int, default=3,
                    help='untimed warm-up iterations of the CPU restatement (BASELINE.md: 3)')
    ap.add_argument('--cpu-budget-s', type=float, default=90.0,
                    help='wall-clock bound of the cpu_baseline leg: on a host whose first iteration '
                         'says 3 + 10 would not fit, the counts shrink (and `sample` says so)')
    ap.add_argument('--no-alt-plan', action='store_true',
                    help='skip the extra measurements of the same workload under the exact-split '
                         '(fp32x3) and the fp32-MFMA-only plans')
    ap.add_argument('--alt-steps', type=int, default=40)
    ap.add_argument('--no-parity-check', action='store_true',
                    help='skip the in-run parity block (step-0 losses / logits of the headline '
                         'plan against the fp32-MFMA and exact-split plans, on the bench inputs '
                         'and on skewed statistics); without it the line is not self-certifying')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='skip the configs[3] (C = 80, bf16 MFMA operands) and configs[4] (TTA '
                         'inference) measurements appended to the default one-GPU line')
    ap.add_argument('--allreduce-chunks', type=int, default=0, help='0 = auto (engine.py)')
    ap.add_argument('--no-conv-streams', action='store_true',
                    help='one launch per conv layer for all images instead of one stream per image')
    ap.add_argument('--mfma-dtype', default='fp16x2', choices=['fp16x2', 'fp32x3', 'fp32', 'bf16'],
                    help="fp16x2 (default), fp32x3 and fp32 are all fp32 arithmetic (BASELINE "
                         "configs[1]/[2]): fp32 = v_mfma_f32_32x32x2_f32 everywhere; fp32x3 = fc6/fc7 "
                         "GEMMs as exact 3-way bf16 splits, 6 bf16-MFMA passes; fp16x2 = the same "
                         "GEMMs as row-scaled 2-way f16 splits, 3 f16-MFMA passes; both accumulate "
                         "in fp32 and are as accurate as the fp32 MFMA (tests/test_gpu_x3.py, "
                         "test_gpu_h2.py).  bf16 = the configs[3] option (operands rounded to bf16, "
                         "fp32 storage + loss)")
    ap.add_argument('--cpu-rois', type=int, default=500)
    ap.add_argument('--infer', action='store_true',
                    help='BASELINE configs[4] instead of the training step: inference with the '
                         'yaml\'s 10-pass multi-scale + flip TTA (scales 480/576/688/864/1200) on a '
                         'synthetic image with --infer-rois proposals; a step = one image end to '
                         'end (device image prep, dedup, 10 forward passes, TTA mean, NMS, top-100)')
    ap.add_argument('--infer-rois', type=int, default=4000)
    ap.add_argument('--infer-height', type=int, default=375)
    ap.add_argument('--infer-width', type=int, default=500)
    return ap.parse_args()


def infer_main(args):
    """python bench.py --infer: ms per image of configs[4] on one GPU (no data-parallel path:
    images are independent; N GPUs = N replicas)."""
    import torch
    torch.cuda.set_device(0)
    emit(infer_measure(args, torch.device('cuda', 0), max(args.steps // 4, 4),
                       max(args.warmup // 4, 2)))


def infer_measure(args, dev, steps, warmup):
    """configs[4]: `steps` synthetic images through the yaml's 10-pass TTA end to end, the fc6
    forward launches of every pass timed with HIP events (engine.timing_events)."""
    import numpy as np
    import torch
    from detectron.core import config as c
    from detectron.core import test_wsl
    from detectron.core.executor import NetExecutor
    from detectron.datasets import synthetic
    from detectron.roi_data.minibatch_wsl import get_im_scale
    import detectron.modeling.model_builder_wsl as mb
    c.reset_cfg()
    c.merge_cfg_from_file(os.path.join(ROOT, 'na-fwebsod_amd', 'configs', 'flickr_voc',
                                       'na_wsddn_V-16-C5_1x.yaml'))
    c.merge_cfg_from_list(['NUM_GPUS', 1, 'TEST.BBOX_AUG.ENABLED', True,
                           'NAWS.MFMA_DTYPE', args.mfma_dtype])
    cfg = c.cfg
    num_fg = cfg.MODEL.NUM_CLASSES - 1
    model = mb.create(cfg.MODEL.TYPE, train=False)
    ex = NetExecutor(model, dev)
    ex.load_blobs(synthetic.init_blobs(num_fg, seed=11))
    h, w = args.infer_height, args.infer_width
    entries = synthetic.make_roidb(4, args.infer_rois, num_fg, h, w, seed=11)
    ims = [np.random.default_rng(e['seed']).integers(0, 256, (h, w, 3), dtype=np.uint8)
           for e in entries]
    assert test_wsl.device_post_supported(ex, ims[0])

    def run(k):
        n = 0
        for i in range(k):
            e = entries[i % len(entries)]
            cls_boxes = test_wsl.im_detect_all(ex, ims[i % len(ims)], e['boxes'], e['obn_scores'])
            n += sum(len(b) for b in cls_boxes[1:])
        torch.cuda.synchronize()
        return n
    run(warmup)
    ev = []
    eng = getattr(ex, 'engine', None)
    if eng is not None:
        eng.timing_events = ev
    t0 = time.perf_counter()
    ndet = run(steps)
    dt = time.perf_counter() - t0
    if eng is not None:
        eng.timing_events = None
    fc6_ms = sum(a.elapsed_time(b) for a, b in ev) / steps if ev else None
    passes = test_wsl.tta_passes()
    # algorithmic work per image (conv 463.7 GFLOP at 600x1000, scaled by the
    # input area; one head branch: fc6 + fc7 + fc8 on the proposals that survive the dedup)
    conv, head = 0.0, 0.0
    for s, m, _f in passes:
        sc = get_im_scale((h, w), s, m)
        conv += 463.7e9 * (round(h * sc) * round(w * sc)) / 600000.0
        head += 2.0 * args.infer_rois * 4096 * (25088 + 4096 + 2 * num_fg)
    tf = (conv + head) / (dt / steps) / 1e12
    peak = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1) if args.mfma_dtype == 'fp16x2' else FP32_MFMA_PEAK_TFLOPS
    res = {
        'metric': 'inference images/sec, %d-pass multi-scale+flip TTA, %d proposals '
                  '(BASELINE configs[4])' % (len(passes), args.infer_rois),
        'value': round(steps / dt, 3), 'unit': 'images/sec', 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (2xf16 split, 3-pass f16 MFMA, fp32 accumulate)' if args.mfma_dtype == 'fp16x2'
                 else args.mfma_dtype,
        'data': 'synthetic',
        'config': {'workload': 'configs[4] flickr_voc na_wsddn TTA inference: image %dx%d, scales '
                               '480/576/688/864/1200 +flip, %d proposals' % (h, w, args.infer_rois),
                   'passes': len(passes), 'detections_per_image': round(ndet / steps, 1),
                   'device_post': bool(cfg.NAWS.DEVICE_POST)},
        'roofline': {'bound': 'mfma', 'kernel': 'whole image: conv bodies + fc6/fc7 of all passes',
                     'achieved': round(tf, 1), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(tf / peak, 4), 'traffic': None,
                     'algorithmic_gflop_per_image': round((conv + head) / 1e9, 1),
                     # the dominant kernel of the path: the fc6 forward GEMM of the clean branch
                     # (M = surviving proposals, N = 4096, K = 25088), summed over the passes
                     'fc6_fwd_ms_per_image': None if fc6_ms is None else round(fc6_ms, 3),
                     'fc6_fwd_launches_per_image': round(len(ev) / steps, 1) if ev else None},
    }
    if fc6_ms:
        f6 = len(passes) * 2.0 * args.infer_rois * 4096 * 25088 / (fc6_ms * 1e-3) / 1e12
        res['roofline'].update(fc6_fwd_tflops=round(f6, 1), fc6_fwd_frac=round(f6 / peak, 4))
    del ex, model
    torch.cuda.empty_cache()
    return res


def host_cpu_limits():
    """What bounds this process's CPU use: logical CPUs, the scheduler affinity mask, and the
    cgroup CPU quota (v2 cpu.max, else v1 cfs quota / period), in cores; None = unlimited."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota = None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    return {'logical_cpus': os.cpu_count() or 1, 'affinity_cpus': aff,
            'cgroup_cpu_quota': None if quota is None else round(quota, 2)}


def pick_cpu_threads(limits):
    """Thread count for the torch-CPU stages of the baseline: a short probe (one conv3-class 3x3
    convolution, 44 GFLOP, and one fc6-class product, 103 GFLOP) under each candidate count, the
    fastest wins.  One image through 13 conv layers does not scale to every hardware thread of a
    two-socket host: round 4's line ran all 256 and reached 0.1 TFLOP/s on the conv stage."""
    import torch
    cap = limits['affinity_cpus']
    if limits['cgroup_cpu_quota']:
        cap = max(1, min(cap, int(limits['cgroup_cpu_quota'] + 0.5)))
    cands = sorted({c for c in (cap, cap // 2, cap // 4, 64, 32, 16) if 1 <= c <= cap})
    x = torch.randn(1, 256, 150, 250)
    w = torch.randn(256, 256, 3, 3)
    a = torch.randn(500, 25088)
    b = torch.randn(4096, 25088)
    table = {}
    for n in cands:
        torch.set_num_threads(n)
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            torch.nn.functional.conv2d(x, w, padding=1)
            t1 = time.perf_counter()
            torch.mm(a, b.t())
            t2 = time.perf_counter()
            if rep:
                best = (t1 - t0, t2 - t1)
        table[n] = best
    pick = min(table, key=lambda n: sum(table[n]))
    torch.set_num_threads(pick)
    gf = {str(n): [round(44.2 / c / 1e3, 3), round(102.8 / f / 1e3, 3)] for n, (c, f) in table.items()}
    return pick, gf


def cpu_baseline(args, num_fg):
    """CPU restatement of the reference path (oracle/, torch-CPU + C ops) on BASELINE
    configs[0]: 2 synthetic 600x1000 images x 500 proposals; a thread-count probe, --cpu-warmup
    (3) warm-up iterations, then --cpu-iters (10) timed fwd+bwd+SGD iterations with per-stage wall
    times - BASELINE.md section 3's 3 + 10, shrunk only if the host cannot fit them into
    --cpu-budget-s (and `sample` then says so)."""
    import numpy as np
    import torch
    from detectron.datasets import synthetic
    from oracle import oracle
    limits = host_cpu_limits()
    threads, probe = pick_cpu_threads(limits)
    roidb = synthetic.make_roidb(2, args.cpu_rois, num_fg, args.height, args.width, seed=11)
    mb = synthetic.make_minibatch(roidb, num_fg)
    blobs = synthetic.init_blobs(num_fg, seed=11)
    rt = mb['rois'].shape[0]
    rng = np.random.default_rng(0)
    masks = {k: (rng.uniform(size=(rt, 4096)) > 0.5).astype(np.float32)
             for k in ('drop6', 'drop7', '_[noisy]_drop6', '_[noisy]_drop7')}
    lr = np.array([1e-3], np.float32)
    state = {}

    def iteration(stages):
        ref = oracle.full_forward_backward(blobs, mb, masks, num_fg, timings=stages)
        t0 = time.perf_counter()
        for name, g in ref['grads'].items():
            p = blobs[name].numpy().reshape(-1)
            if name not in state:
                state[name] = (np.zeros_like(p), np.zeros_like(p))
            m, a = state[name]
            bias = name.endswith('_b')
            oracle.acm_sgd(np.ascontiguousarray(g.reshape(-1)), m, lr, p, a, 0.9, 0,
                           0.0 if bias else 5e-4, 1, 2, 2.0 if bias else 1.0, 1)
        stages['sgd'] = stages.get('sgd', 0.0) + time.perf_counter() - t0

    # 3 warm-up + 10 timed iterations (~2.2 s each on the GPU box's 16
    # quota cores: ~30 s).  The first iteration (thread pool, oneDNN primitives, page faults)
    # sizes the rest against --cpu-budget-s so that a slow host still finishes within minutes.
    want_warm, want_iters = max(1, args.cpu_warmup), max(1, args.cpu_iters)
    t0 = time.perf_counter()
    iteration({})
    first = time.perf_counter() - t0
    fit = max(2, int(args.cpu_budget_s / max(first, 1e-3)))      # iterations the budget holds
    warm = min(want_warm, max(1, fit // 4))
    iters = min(want_iters, max(1, fit - warm))
    for _ in range(warm - 1):
        iteration({})
    stages = {}
    per_iter = []
    for _ in range(iters):
        t0 = time.perf_counter()
        iteration(stages)
        per_iter.append(time.perf_counter() - t0)
    dt = sum(per_iter) / iters
    bounded = (warm, iters) != (want_warm, want_iters)
    conv_tf = 2 * 463.7 / (stages.get('conv', 0.0) / iters) / 1e3 if stages.get('conv') else None
    res = {'value': round(2.0 / dt, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
           'sample': '%d warm-up + %d timed iterations%s of fwd+bwd+SGD on 2 images %dx%d x %d '
                     'proposals (BASELINE configs[0]), fp32, torch-CPU conv/fc on %d threads + '
                     'single-thread C oracle ops, %.2f s per iteration (min %.2f, max %.2f)'
                     % (warm, iters,
                        ' (BASELINE.md asks %d + %d; shrunk to fit --cpu-budget-s %.0f: the first '
                        'iteration took %.1f s)' % (want_warm, want_iters, args.cpu_budget_s, first)
                        if bounded else ' (BASELINE.md section 3)',
                        args.height, args.width, args.cpu_rois, threads, dt, min(per_iter),
                        max(per_iter)),
           'warmup_iterations': warm, 'timed_iterations': iters,
           'ms_per_iteration': round(dt * 1e3, 1),
           'threads_probe_tflops_conv_fc': probe}
    res.update(limits)
    if conv_tf is not None:
        res['conv_stage_tflops'] = round(conv_tf, 3)
        if conv_tf < 1.0:
            # what keeps the conv stage under 1 TFLOP/s on this host
            why = []
            if limits['cgroup_cpu_quota'] and limits['cgroup_cpu_quota'] < limits['logical_cpus']:
                why.append('cgroup quota %.1f cores' % limits['cgroup_cpu_quota'])
            if limits['affinity_cpus'] < limits['logical_cpus']:
                why.append('affinity mask %d of %d cpus' % (limits['affinity_cpus'], limits['logical_cpus']))
            why.append('batch-1 3x3 convolutions: the best probed thread count (%d) reaches %.2f '
                       'TFLOP/s on a conv3-class layer' % (threads, probe[str(threads)][0]))
            res['conv_stage_limit'] = '; '.join(why)
    for k, v in stages.items():
        res['stage_ms_' + k] = round(v / iters * 1e3, 1)
    return res


def step0_probe(out):
    """What the in-run parity check compares: the per-image losses and the fc8 logits
    (fc8c | fc8d | noisy_fc8c | noisy_fc8d) of a plan's FIRST step - same inputs, same initial
    weights, same counter-based dropout masks in every plan."""
    return {k: out[k].detach().double().cpu().numpy() for k in ('loss_cls', 'loss_cls_noise', 'logits')}


def parity_against(base, other):
    """(max relative loss difference over images and both losses, max logit difference /
    max|logit|) of `base` against `other`."""
    import numpy as np
    lr = max(float(np.abs(base[k] - other[k]).max() / np.abs(other[k]).max())
             for k in ('loss_cls', 'loss_cls_noise'))
    lg = float(np.abs(base['logits'] - other['logits']).max() / np.abs(other['logits']).max())
    return lr, lg


def alt_plan(args, dev, num_fg, B, t, seg, mode, blobs=None, steps=None):
    """The same workload under another arithmetic plan, with its own roofline block for the
    fc6-forward launch (HIP events on the launch stream, as for the headline plan).  steps = 0:
    only the first step's probe (the in-run parity check on other statistics)."""
    import torch
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    eng = WsddnEngine(num_fg + 1, dev, dilation=2, dropout=0.5, is_mean=True, momentum=0.9,
                      weight_decay=5e-4, iter_size=1, gpu_num=B, seed=11, mfma_dtype=mode)
    if blobs is None:
        blobs = synthetic.init_blobs(num_fg, seed=11)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    del blobs
    eng.set_lr(args.lr)
    steps = max(1, args.alt_steps) if steps is None else steps
    warm = 3
    ev, pev = [], []
    probe = []

    def run(n, timed):
        eng.timing_events = ev if timed else None
        eng.phase_events = pev if timed else None
        for _ in range(n):
            # (= forward_backward + sgd_step; one rank, split plans: fc6_w updated in its wgrad GEMM)
            out = eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            if not probe:
                probe.append(step0_probe(out))
        eng.flush()
        torch.cuda.synchronize()
        return out
    if steps == 0:
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg,
                                   compute_grads=False)
        torch.cuda.synchronize()
        del eng
        torch.cuda.empty_cache()
        return {'mfma_dtype': mode, 'probe': step0_probe(out)}
    run(warm, False)
    t0 = time.perf_counter()
    out = run(steps, True)
    dt = time.perf_counter() - t0
    rt = t['rois'].shape[0]
    kern_ms = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
    conv_ms = [e0.elapsed_time(e1) for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:])
               if n1 == 'conv_body']
    flops = 2.0 * rt * 8192 * 25088
    achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
    peak, kname = plan_peak_and_kernel(mode)
    roof = {'bound': 'mfma', 'kernel': 'fc6 fwd (both branches, M=%d N=8192 K=25088): %s' % (rt, kname),
            'kernel_ms': round(kern_ms, 4), 'achieved': round(achieved, 2) if achieved else None,
            'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4) if achieved else None,
            'traffic': None}
    if conv_ms:
        roof['conv_stack_ms'] = round(sum(conv_ms) / len(conv_ms), 3)
    # (only the plans whose PMC passes are part of the round's profile set: the strict plans'
    # last passes are round 1's, of kernels that have since changed)
    tjd, tsrc = committed_traffic(mode) if mode in ('bf16', 'fp16x2') else (None, None)
    if tjd is not None and (args.height, args.width, args.rois, B) == (600, 1000, 2000, 2):
        roof.update(traffic=tjd['hbm_bytes_per_launch'], traffic_source=tsrc,
                    traffic_measured_in_run=False,
                    traffic_ratio_vs_algorithmic=round(tjd['hbm_bytes_per_launch'] /
                                                       tjd['algorithmic_bytes_per_launch'], 3))
    del eng
    torch.cuda.empty_cache()
    return {'mfma_dtype': mode, 'value': round(B * steps / dt, 3), 'unit': 'images/sec',
            'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'warmup': warm,
            'final_loss': round(float(out['loss_cls'].sum().item() +
                                      out['loss_cls_noise'].sum().item()), 5),
            'roofline': roof, 'probe': probe[0]}


PARITY_LOSS_TOL = 1e-4      # north_star: loss parity <= 1e-4 relative
PARITY_LOGIT_TOL = 1e-5     # logits: max difference / max|logit| (north_star asks 1e-4)


def parity_in_run(args, dev, num_fg, B, t, seg, base, res):
    """The headline plan certifies itself against the strict plans INSIDE the driver's run:
    step-0 losses and fc8 logits (same inputs, initial weights and dropout masks) of the 2 x f16
    split plan against (a) the plan with every GEMM on the fp32 MFMA and (b) the exact 3 x bf16
    split, both on the bench inputs and on skewed statistics (synthetic.skew_blobs / skew_images:
    per-channel weight scales log-uniform over 2^+-6, non-zero biases, a third of each image at
    2^-12).  `ok` False -> bench.py exits non-zero after printing the line."""
    import torch
    from detectron.datasets import synthetic
    blk = {'loss_tol': PARITY_LOSS_TOL, 'logit_tol': PARITY_LOGIT_TOL}
    worst_l, worst_g = 0.0, 0.0
    for key, tag in (('fp32_mfma_plan', 'fp32_mfma'), ('fp32x3_plan', 'fp32x3')):
        lr, lg = parity_against(base, res[key]['probe'])
        blk['loss_rel_vs_' + tag], blk['logit_maxrel_vs_' + tag] = float('%.3g' % lr), float('%.3g' % lg)
        worst_l, worst_g = max(worst_l, lr), max(worst_g, lg)
    # the same check on non-Kaiming statistics (forward only, one step per plan)
    blobs = synthetic.skew_blobs(synthetic.init_blobs(num_fg, seed=11), seed=11)
    ts = dict(t)
    ts['data'] = torch.from_numpy(synthetic.skew_images(t['data'].cpu().numpy())).to(dev)
    sk = {m: alt_plan(args, dev, num_fg, B, ts, seg, m, blobs=blobs, steps=0)['probe']
          for m in ('fp16x2', 'fp32', 'fp32x3')}
    for m, tag in (('fp32', 'fp32_mfma'), ('fp32x3', 'fp32x3')):
        lr, lg = parity_against(sk['fp16x2'], sk[m])
        blk['skewed_loss_rel_vs_' + tag] = float('%.3g' % lr)
        blk['skewed_logit_maxrel_vs_' + tag] = float('%.3g' % lg)
        worst_l, worst_g = max(worst_l, lr), max(worst_g, lg)
    # the yardstick: how far the two strict plans (both exact fp32 products, different
    # summation orders) are from each other on the same inputs
    lr, lg = parity_against(res['fp32x3_plan']['probe'], res['fp32_mfma_plan']['probe'])
    blk['fp32x3_vs_fp32_mfma_loss_rel'], blk['fp32x3_vs_fp32_mfma_logit_maxrel'] = \
        float('%.3g' % lr), float('%.3g' % lg)
    blk['ok'] = bool(worst_l <= PARITY_LOSS_TOL and worst_g <= PARITY_LOGIT_TOL)
    return blk


def extra_configs(args, dev, B, res, cfg, roof):
    """BASELINE configs[3] (C = 80, bf16 MFMA operands, fp32 loss; its 1-GPU share) and configs[4]
    (TTA inference) measured in the same driver run, each with its fc6-forward roofline; flat
    copies under config / roofline."""
    import numpy as np
    import torch
    from detectron.datasets import synthetic
    c80 = 80
    mb = synthetic.make_minibatch(synthetic.make_roidb(B, args.rois, c80, args.height, args.width,
                                                       seed=11), c80)
    t80 = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)).tolist()
    r = alt_plan(args, dev, c80, B, t80, seg, 'bf16')
    r.pop('probe', None)
    r['workload'] = ('configs[3] flickr_coco na_wsddn C=80 (1-GPU share): %d img %dx%d x %d rois, '
                     'bf16 MFMA conv/fc6/fc7 operands, fp32 storage/fc8/loss/SGD'
                     % (B, args.height, args.width, args.rois))
    res['bf16_c80_plan'] = r
    cfg['bf16_c80_plan_images_per_sec'] = r['value']
    cfg['bf16_c80_plan_ms_per_step'] = r['ms_per_step']
    for k in ('kernel_ms', 'achieved', 'peak', 'frac', 'traffic', 'traffic_ratio_vs_algorithmic'):
        roof['bf16_c80_plan_fc6_fwd_' + {'achieved': 'tflops'}.get(k, k)] = r['roofline'].get(k)
    roof['bf16_c80_plan_conv_stack_ms'] = r['roofline'].get('conv_stack_ms')
    del t80
    torch.cuda.empty_cache()
    ia = argparse.Namespace(**vars(args))
    ia.mfma_dtype = 'fp16x2'
    ti = infer_measure(ia, dev, steps=16, warmup=4)
    res['tta_infer'] = {k: ti[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps',
                                            'warmup', 'dtype', 'config', 'roofline')}
    cfg['tta_infer_ms_per_image'] = ti['ms_per_step']
    cfg['tta_infer_passes'] = ti['config']['passes']
    cfg['tta_infer_detections_per_image'] = ti['config']['detections_per_image']
    for k in ('achieved', 'peak', 'frac', 'fc6_fwd_ms_per_image', 'fc6_fwd_tflops', 'fc6_fwd_frac'):
        roof['tta_infer_' + {'achieved': 'whole_image_tflops'}.get(k, k)] = ti['roofline'].get(k)


def project_n_ranks(eng, t, seg, n, cus, gbps, steps, sharded=False, pipeline=True):
    """ms/step of the N-rank schedule on this one GPU with reducer.EmulatedExchange in the
    all-reduce's place (restores the engine's own reducer afterwards).  sharded: the
    NAWS.SHARDED_UPDATE schedule, this process playing rank 0 of N (it updates 1 / N of fc6_w's
    rows; the rest keep their values - timing only)."""
    import torch
    from naws_hip.reducer import EmulatedExchange
    eng.flush()
    saved = (eng.reducer, eng.allreduce_chunks, eng.phase_events, eng.timing_events,
             eng.update_events, eng.comm_events, eng.pipeline_update)
    eng.pipeline_update = bool(pipeline)
    ex = EmulatedExchange(eng.device, n, cus, gbps)
    ex.log = []
    eng.reducer, eng.allreduce_chunks = ex, (4 if n == 2 else 2)
    eng.sharded_update, eng._shard = bool(sharded), None
    eng.timing_events = eng.update_events = eng.comm_events = None
    pev = []
    try:
        for it in range(5 + steps):
            if it == 5:
                eng.flush()
                torch.cuda.synchronize()
                eng.phase_events = pev
                t0 = time.perf_counter()
            if pipeline and not sharded:
                # (as the training loop does: train_step queues the update's parts from inside
                # backward, each right behind the message it waits for)
                eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            else:
                eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
                eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        eng.flush()
        (eng.reducer, eng.allreduce_chunks, eng.phase_events, eng.timing_events,
         eng.update_events, eng.comm_events, eng.pipeline_update) = saved
        eng.sharded_update, eng._shard, eng._mom_synced = False, None, True
        eng._pipe_sent = False
    exposed = [e0.elapsed_time(e1) for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:]) if n1 == 'join_update']
    # (pipelined: the head joins the update piece by piece INSIDE its forward - what is exposed
    # there shows as a longer head_fwd stage)
    hf = [e0.elapsed_time(e1) for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:]) if n1 == 'head_fwd']
    st = {}
    for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:]):
        if n1 != 'start':
            st.setdefault(n1, []).append(e0.elapsed_time(e1))
    between = [e0.elapsed_time(e1) for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:]) if n1 == 'start']
    return {'ms_per_step': round(ms, 3), 'cus': ex.cus, 'gbps': ex.gbps, 'chunks': 4 if n == 2 else 2,
            'pipelined': bool(pipeline and not sharded), 'head_fwd_ms': sum(hf) / max(len(hf), 1),
            'stages': {k: round(sum(v) / len(v), 3) for k, v in st.items()},
            'between_steps_ms': round(sum(between) / max(len(between), 1), 3),
            'messages_per_step': ex.log[:len(ex.log) // (5 + steps)],
            'bytes_per_step': ex.total_bytes / (5 + steps),
            'exposed_ms': sum(exposed) / max(len(exposed), 1)}


def committed_traffic(mode):
    """(the newest profiles/r*_bench*traffic*.json of this arithmetic plan, its path) or (None,
    None): HBM-side traffic of the plan's fc6-forward launch from the rocprofv3 PMC passes of
    this same command (tools/profile_bench.sh + summarize_profile.py); a run never re-measures it."""
    import glob
    tj = [f for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench*traffic*.json')))
          if json.load(open(f)).get('mfma_dtype', 'fp32') == mode]
    if not tj:
        return None, None
    return json.load(open(tj[-1])), os.path.relpath(tj[-1], ROOT)


def plan_peak_and_kernel(mode):
    """(ceiling in ALGORITHMIC TFLOP/s, name) of a plan's fc6-forward kernel.  The split plans
    execute 3 (fp16x2) / 6 (fp32x3) 16-bit MFMA flops per algorithmic fp32 flop, so their
    ceiling is the dense 16-bit peak divided by the pass count - a derived figure, not a
    hardware number; fp32 = the v_mfma_f32_32x32x2_f32 datasheet peak."""
    if mode == 'bf16':
        return BF16_MFMA_PEAK_TFLOPS, ('gemm_x3_m16_kernel<256,256,4x2 waves,2 stages,1 plane x 4 '
                                       'K-slabs> (bf16 slab operands, v_mfma_f32_16x16x32_bf16)')
    if mode == 'fp32x3':
        return round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), X3_KERNEL_NAME
    if mode == 'fp16x2':
        return round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1), ('gemm_x3_m16_kernel<256,256,4x2 waves,2 stages,2 '
                                                       'planes x 2 K-slabs,f16> = 3 x '
                                                       'v_mfma_f32_16x16x32_f16 per fp32 product')
    return FP32_MFMA_PEAK_TFLOPS, 'gemm_f32_kernel<256,256,16,KC,KC,4x4 waves> (v_mfma_f32_32x32x2_f32)'


X3_KERNEL_NAME = ('gemm_x3_m16_kernel<256,128,4x2 waves,2 stages,3 planes x 2 K-slabs> = 6 x '
                  'v_mfma_f32_16x16x32_bf16 per fp32 product')


# ---------------------------------------------------------------------------------------------
# N > 1 hardening: the same five functions serve the GPU job (EngineJob) and
# the --dry-run job on CPU tensors over gloo (DryJob), so the CPU tests exercise the very code
# the first multi-GPU RCCL run will execute.
# ---------------------------------------------------------------------------------------------
def ranks_agree(job, pg, rank, world):
    """(True when every rank holds bit-identical state, [names of the buffers that differ]).
    COLLECTIVE: naws_hip.reducer.ranks_agree over the job's state buffers."""
    from naws_hip.reducer import ranks_agree as _agree
    return _agree(job.state_tensors(), pg, rank, world)


def bare_allreduce(job, pg, world, iters=5, warm=2):
    """The gradient exchange ALONE: the step's own messages (job.message_slices(): fc6_w's
    gradient in its row chunks, then the small gradients - the same slices of the same arena),
    handed to the collective back to back with no compute beside them.  -> (ms per exchange,
    bus bandwidth in GB/s = 2 (N-1)/N x bytes / time, bytes)."""
    import torch.distributed as dist
    slices = job.message_slices()
    nbytes = sum(int(x.numel()) * x.element_size() for x in slices)

    def once():
        works = [dist.all_reduce(x, group=pg, async_op=True) for x in slices]
        for w in works:
            w.wait()
        job.sync()
    for _ in range(warm):
        once()
    dist.barrier(group=pg)
    job.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    dist.barrier(group=pg)
    job.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    bus = 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9 if world > 1 else 0.0
    for x in slices:
        x.zero_()                   # (sums of zeros stay zeros; a dry-run arena is reset too)
    return ms, bus, nbytes


def timed_steps(job, pg, rank, world, steps, warm=0):
    """`warm` untimed + `steps` timed steps of the job's CURRENT route between barrier + device
    sync on both sides; -> (seconds of the slowest rank, [seconds per rank]).  COLLECTIVE."""
    import torch
    import torch.distributed as dist
    for _ in range(warm):
        job.step(False)
    job.flush()
    if pg is not None:
        dist.barrier(group=pg)
    job.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step(True)
    job.flush()          # the last iteration's (deferred) all-reduce + SGD belongs to the K steps
    if pg is not None:
        dist.barrier(group=pg)
    job.sync()
    dt = time.perf_counter() - t0
    rank_dt = [dt]
    if pg is not None:
        td = torch.zeros((world,), device=job.device, dtype=torch.float64)
        td[rank] = dt
        dist.all_reduce(td, group=pg)
        rank_dt = td.cpu().tolist()
    return max(rank_dt), rank_dt


def route_ab(job, pg, rank, world, steps, hb, images_per_step):
    """After the timed region, in the SAME job: K steps of the other update routes, each between
    its own barriers, each followed by the rank-digest check.  -> flat keys for the line.
    `value` stays the default route's; a leg that cannot run says why."""
    from naws_hip.reducer import owner_blocks
    out = {}
    was = job.route()
    legs = []
    if was['sharded']:
        legs = [('unpipelined', dict(pipeline_update=False, sharded_update=False))]
    else:
        other = not was['pipelined']
        legs.append(('pipelined' if other else 'unpipelined',
                     dict(pipeline_update=None if other else False, sharded_update=False)))
        # the plainest exchange there is - fc6_w's gradient as ONE message behind the whole wgrad,
        # one update launch - as the yardstick of what chunking + pipelining buy on real links
        legs.append(('one_message', dict(pipeline_update=False, sharded_update=False, chunks=1)))
        if owner_blocks(job.rows6, world) is not None and job.can_shard:
            legs.append(('sharded', dict(pipeline_update=False, sharded_update=True)))
        else:
            out['value_sharded'] = None
            out['sharded_skipped'] = 'fc6_w rows do not divide into 32-row blocks over %d ranks' % world
    for name, route in legs:
        hb.phase('ab_' + name, steps=steps + 3)
        job.set_route(**route)
        got = job.route()
        if route.get('chunks'):
            out['chunks_' + name] = got['chunks']
        if (name == 'pipelined') != got['pipelined'] or (name == 'sharded') != got['sharded']:
            out['value_' + name] = None          # (the plan cannot take that route: said, not faked)
            out[name + '_skipped'] = 'route not available on this plan'
            continue
        dt, _ = timed_steps(job, pg, rank, world, steps, warm=3)
        out['value_' + name] = round(world * images_per_step * steps / dt, 3)
        out['ms_per_step_' + name] = round(dt / steps * 1e3, 3)
        ok, bad = ranks_agree(job_after_gather(job), pg, rank, world)
        out['rank_digest_equal_' + name] = bool(ok)
        if not ok:
            out['rank_digest_differs_' + name] = ','.join(bad)
    job.set_route(pipeline_update=None if was['pipelined'] else False, sharded_update=was['sharded'],
                  chunks=was['chunks'])
    return out


def job_after_gather(job):
    job.gather_sharded_state()       # (COLLECTIVE; no-op off the sharded route)
    return job


class EngineJob(object):
    """WsddnEngine + its resident inputs behind the few calls the N > 1 skeleton needs."""

    def __init__(self, eng, t, seg, two_call=False):
        self.eng, self.t, self.seg, self.two_call = eng, t, seg, two_call
        self.device = eng.device
        self.rows6 = 8192
        self.can_shard = eng.mfma_dtype == 'fp16x2' and eng.iter_size == 1
        self.timed_hooks = None       # (ev, pev, uev, cev) lists while the headline region runs
        self.last = None
        self.probe_fn, self.first_probe = None, None

    def step(self, timed):
        e, t = self.eng, self.t
        hooks = self.timed_hooks if timed and self.timed_hooks else (None, None, None, None)
        e.timing_events, e.phase_events, e.update_events, e.comm_events = hooks
        if self.two_call:
            out = e.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=self.seg)
            e.sgd_step()
        else:       # the same two calls; without a gradient exchange fc6_w is updated by its wgrad GEMM
            out = e.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=self.seg)
        self.last = out
        if self.first_probe is None and self.probe_fn is not None:
            self.first_probe = self.probe_fn(out)
        return out

    def flush(self):
        self.eng.flush()

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def state_tensors(self):
        return self.eng.state_tensors()

    def gather_sharded_state(self):
        self.eng.gather_sharded_state()

    def broadcast_parameters(self):
        self.eng.broadcast_parameters(0)

    def message_slices(self):
        from naws_hip.reducer import message_plan, message_slice
        e = self.eng
        plan = message_plan(e.arena, 2 * 4096, e.allreduce_chunks, True, False)
        return [message_slice(e.arena, e.grads, k, r, e.k6) for k, r in plan]

    def route(self):
        e = self.eng
        return dict(pipelined=bool(e._pipelined()), sharded=bool(e._shard_blocks() is not None),
                    chunks=int(e.allreduce_chunks))

    def set_route(self, pipeline_update=None, sharded_update=False, chunks=None):
        self.eng.set_update_route(pipeline_update, sharded_update)
        if chunks:
            self.eng.allreduce_chunks = int(chunks)     # (read per backward: engine.message_plan)


class DryJob(object):
    """--dry-run: the N > 1 skeleton on a small CPU parameter arena over gloo - real message
    plans, the real ArenaReducer, a trivial deterministic 'gradient' - so that broadcast, rank
    digests, the bare exchange, the route A/B and the supervisor's fallbacks run on a CPU box."""

    def __init__(self, pg, rank, world, pipelined, sharded, chunks):
        import torch
        from naws_hip.engine import ParamArena
        from naws_hip.reducer import ArenaReducer
        self.device = torch.device('cpu')
        self.pg, self.rank, self.world = pg, rank, world
        self.rows6, self.k6, self.can_shard = 256, 48, True
        h, c = 128, 5
        specs = [('fc6_w', (h, self.k6)), ('_[noisy]_fc6_w', (h, self.k6)), ('fc6_b', (h,)),
                 ('_[noisy]_fc6_b', (h,)), ('fc7_w', (h, h)), ('_[noisy]_fc7_w', (h, h)),
                 ('fc7_b', (h,)), ('_[noisy]_fc7_b', (h,)), ('fc8c_w', (c, h)), ('fc8d_w', (c, h)),
                 ('noisy_fc8c_w', (c, h)), ('noisy_fc8d_w', (c, h)), ('fc8c_b', (c,)),
                 ('fc8d_b', (c,)), ('noisy_fc8c_b', (c,)), ('noisy_fc8d_b', (c,))]
        self.arena = ParamArena(specs, self.device)
        g = torch.Generator().manual_seed(11 + 100 * rank)   # every rank starts DIFFERENT
        self.params = torch.randn((self.arena.total,), generator=g)
        self.momentum = torch.randn((self.arena.total,), generator=g)
        self.grads = torch.zeros((self.arena.total,))
        self.reducer = ArenaReducer(pg, world)
        self.reducer.force = pg is not None
        self.pipelined, self.sharded, self.chunks = bool(pipelined), bool(sharded), int(chunks)
        self.it = 0
        self.inject = os.environ.get('NAWS_BENCH_INJECT', '')

    def _inject(self, what):
        """NAWS_BENCH_INJECT='<what>:<route>:<rank>[:<step>]' - the tests' stall / death / divergence
        on one rank of one route (what in stall, die, diverge)."""
        p = self.inject.split(':')
        if len(p) < 3 or p[0] != what or int(p[2]) != self.rank:
            return False
        route = 'sharded' if self.sharded else 'pipelined' if self.pipelined else 'unpipelined'
        return p[1] in (route, 'any') and self.it >= int(p[3] if len(p) > 3 else 0)

    def step(self, timed):
        import torch
        from naws_hip.reducer import message_plan, message_slice, owner_blocks, owner_pieces
        if self._inject('stall'):
            time.sleep(3600)
        if self._inject('die'):
            os._exit(9)
        time.sleep(0.001)
        g = torch.Generator().manual_seed(1000 * self.it + self.rank)
        self.grads.copy_(torch.randint(-8, 9, (self.arena.total,), generator=g).float() / 64.0)
        if self._inject('diverge'):
            self.grads[0] += 1.0
        blocks = owner_blocks(self.rows6, self.world) if self.sharded else None
        o6 = self.arena.offsets['fc6_w'][0]
        for kind, rows in message_plan(self.arena, self.rows6, self.chunks, True, self.pipelined):
            sl = message_slice(self.arena, self.grads, kind, rows, self.k6)
            if blocks is not None and kind == 'fc6_w':
                for o, p0, p1 in owner_pieces(rows[0], rows[1], blocks):
                    self.reducer.reduce_to_owner_async(
                        self.grads[o6 + p0 * self.k6:o6 + p1 * self.k6], o)
            else:
                self.reducer.reduce_async(sl)
        self.reducer.wait()
        if self._inject('diverge'):
            self.grads[0] += float(self.rank + 1)     # AFTER the sum: this rank's state walks off
        lr = 1.0 / 1024
        if blocks is None:
            self.momentum.mul_(0.5).add_(self.grads, alpha=lr / self.world)
            self.params.sub_(self.momentum)
        else:       # owner-only update of fc6_w's rows, everything else everywhere, rows gathered
            b0, b1 = blocks[self.rank]
            mine = slice(o6 + b0 * self.k6, o6 + b1 * self.k6)
            rest = slice(o6 + self.rows6 * self.k6, self.arena.total)
            for s_ in (mine, rest):
                self.momentum[s_].mul_(0.5).add_(self.grads[s_], alpha=lr / self.world)
                self.params[s_].sub_(self.momentum[s_])
            self.reducer.gather_blocks_async(self.params[o6:o6 + self.rows6 * self.k6], self.rank)
            self.reducer.wait()
            self._mom_local = True
        self.it += 1

    def flush(self):
        pass

    def sync(self):
        pass

    def state_tensors(self):
        return dict(params=self.params, momentum=self.momentum)

    def gather_sharded_state(self):
        if getattr(self, '_mom_local', False):
            o6 = self.arena.offsets['fc6_w'][0]
            self.reducer.gather_blocks_async(self.momentum[o6:o6 + self.rows6 * self.k6], self.rank)
            self.reducer.wait()
            self._mom_local = False

    def broadcast_parameters(self):
        import torch.distributed as dist
        dist.broadcast(self.params, 0, group=self.pg)
        dist.broadcast(self.momentum, 0, group=self.pg)

    def message_slices(self):
        from naws_hip.reducer import message_plan, message_slice
        return [message_slice(self.arena, self.grads, k, r, self.k6)
                for k, r in message_plan(self.arena, self.rows6, self.chunks, True, False)]

    def route(self):
        return dict(pipelined=self.pipelined and not self.sharded, sharded=self.sharded,
                    chunks=self.chunks)

    def set_route(self, pipeline_update=None, sharded_update=False, chunks=None):
        self.gather_sharded_state()
        self.pipelined = pipeline_update is None or bool(pipeline_update)
        self.sharded = bool(sharded_update)
        if chunks:
            self.chunks = int(chunks)


def init_process_group(backend, rank, world, dev=None):
    """env:// rendezvous (the launcher's store) on the first attempt; a supervisor's fallback
    attempt rendezvouses through a FileStore of its own (NAWS_BENCH_STORE): the launcher's TCP
    store still holds the keys of the attempt that was killed."""
    import torch.distributed as dist
    kw = {}
    if os.environ.get('NAWS_BENCH_STORE'):
        kw['init_method'] = 'file://' + os.environ['NAWS_BENCH_STORE']
    if dev is not None:
        kw['device_id'] = dev
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launcher_command(n, port, argv, python=None, script=None):
    """argv of the child job for a bare `python bench.py --gpus N`: one process per GPU through
    torch.distributed.run on this node (the driver's own N > 1 command line), rendezvous on
    127.0.0.1.  `argv` = this process's arguments after the script name; --self-launch is
    dropped (the children are ranks, not launchers)."""
    rest = [a for a in argv if a != '--self-launch']
    return [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
            '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
            script or os.path.abspath(__file__)] + rest


def launcher_env(environ):
    """Environment of the children: the parent's, minus any rank variables (a stale RANK /
    WORLD_SIZE would be trusted by the ranks), plus HSA_ENABLE_IPC_MODE_LEGACY=0.

    Where that switch comes from: the build environment's own operating notes for this GPU
    pool - the hosts' kernel driver supports only dmabuf IPC, and with the ROCr default
    (legacy IPC handles) any cross-process device-memory sharing, which is how RCCL sets up
    its intra-node xGMI transport, fails with `hipIpcGetMemHandle: invalid argument`.  The
    pool exports the variable itself (in this container and on the GPU boxes); `setdefault`
    only restores it for a caller who scrubbed the environment and never overrides a value
    the operator chose.  It has NOT been exercised here with two RCCL ranks (no multi-GPU
    box was ever available to this project); the one-rank RCCL path (--force-dist) runs with
    it.  Should a site's driver want legacy IPC, exporting HSA_ENABLE_IPC_MODE_LEGACY=1
    before the launch wins."""
    env = {k: v for k, v in environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK',
                        'MASTER_ADDR', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    env['NAWS_BENCH_CHILD'] = '1'
    return env


def is_result_line(line):
    line = line.strip()
    if not (line.startswith('{') and line.endswith('}')):
        return False
    try:
        return 'metric' in json.loads(line)
    except ValueError:
        return False


def self_launch(args, argv):
    """Parent of a bare `python bench.py --gpus N`: it never touches the GPU (a process that has
    initialised HIP must not be replaced, and need not be: the ranks are children).  The
    children's output is passed through as it comes; rank 0's JSON line is held back and printed
    as the LAST line of stdout.  Exit code: the job's."""
    import subprocess
    cmd = launcher_command(args.gpus, free_port(), argv)
    print('bench.py: launching %d rank(s): %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=launcher_env(os.environ), stdout=subprocess.PIPE,
                            stderr=None, text=True, bufsize=1)
    held = None
    for line in proc.stdout:
        if is_result_line(line):
            held = line.strip()
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if held is not None:
        print(held, flush=True)
    if rc != 0:
        sys.exit(rc if rc > 0 else 1)
    if held is None:
        sys.exit('bench.py: the ranks exited 0 without a result line')


def emit(res):
    """The ONE JSON line, as the LAST line of stdout: RCCL writes its version banner through C
    stdio, which would otherwise be flushed at exit, after Python's own line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(res), flush=True)


def n_rank_prologue(job, pg, rank, world, args, hb):
    """What every N > 1 job does before its first step: rank 0's parameters to every rank
    (reference: utils/net_wsl.py:183-207 - the ranks are seeded DIFFERENTLY on purpose, so the
    check below fails without the broadcast), then the gradient exchange alone.  -> keys."""
    out = {}
    hb.phase('setup')
    same_before, _ = ranks_agree(job, pg, rank, world)
    job.broadcast_parameters()
    same_after, bad = ranks_agree(job, pg, rank, world)
    out['params_broadcast_from_rank0'] = True
    out['ranks_equal_before_broadcast'] = bool(same_before)
    if not same_after:
        print('bench.py: ranks differ AFTER the parameter broadcast: %s' % ','.join(bad),
              file=sys.stderr, flush=True)
        sys.exit(EXIT_DIGEST_MISMATCH)
    hb.phase('allreduce_alone')
    share = getattr(args, 'share_gpu', False)
    ms, bus, nbytes = bare_allreduce(job, pg, world, iters=2 if share else 5, warm=1 if share else 2)
    out.update(allreduce_alone_ms=round(ms, 3), allreduce_busbw_GBps=round(bus, 2),
               allreduce_bytes=nbytes)
    return out


EXIT_DIGEST_MISMATCH = 4        # (= naws_hip.supervise.EXIT_DIGEST_MISMATCH)


def check_rank_digests(job, pg, rank, world, hb, where):
    """COLLECTIVE.  Exit code 4 on every rank when the ranks' states differ (under the
    supervisor: fresh workers on the next route of the ladder)."""
    hb.phase('digest')
    ok, bad = ranks_agree(job_after_gather(job), pg, rank, world)
    if not ok:
        if rank == 0:
            print('bench.py: the ranks hold DIFFERENT state %s: %s (route %s)'
                  % (where, ','.join(bad), json.dumps(job.route())), file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(EXIT_DIGEST_MISMATCH)      # (no teardown: a diverged job's collectives may not match)
    return True


def save_main_line(res):
    """Rank 0, as soon as the headline exists: the supervisor prints this file if a LATER leg
    (route A/B, extras) stalls or dies."""
    path = os.environ.get('NAWS_BENCH_MAINLINE')
    if path:
        with open(path + '.tmp', 'w') as f:
            json.dump(res, f)
        os.replace(path + '.tmp', path)


def fallback_keys():
    fb = os.environ.get('NAWS_BENCH_FALLBACK')
    return dict(route_fallback=fb, route_attempt=int(os.environ.get('NAWS_BENCH_ATTEMPT', '0')),
                route_of_attempt=os.environ.get('NAWS_BENCH_ROUTE', 'as launched'),
                supervised=os.environ.get('NAWS_BENCH_WORKER') == '1')


def dry_run(args, rank, world):
    """The rank skeleton of main() without a GPU (tests/test_bench_launcher.py): rendezvous over
    gloo, parameter broadcast, bare exchange, warm-up, rank-digest check, K timed steps between
    barriers, max over ranks, the route A/B legs, ONE JSON line - on DryJob's CPU arena."""
    import torch.distributed as dist
    from naws_hip.supervise import Heartbeat
    hb = Heartbeat()
    hb.phase('init')
    if os.environ.get('NAWS_DRY_RUN_FAIL_RANK') == str(rank):
        sys.exit(3)                      # the launcher test's failing rank
    pg = None
    if world > 1 or args.force_dist:
        pg = init_process_group('gloo', rank, world)
    job = DryJob(pg, rank, world, pipelined=not args.no_pipeline_update and not args.sharded_update,
                 sharded=args.sharded_update, chunks=args.allreduce_chunks or (4 if world == 2 else 2))
    extra = {}
    if pg is not None:
        extra.update(n_rank_prologue(job, pg, rank, world, args, hb))
    hb.phase('warmup', steps=args.warmup)
    for _ in range(args.warmup):
        job.step(False)
    if pg is not None:
        extra['rank_digest_equal'] = check_rank_digests(job, pg, rank, world, hb, 'after warm-up')
    hb.phase('timed', steps=args.steps)
    dt, rank_dt = timed_steps(job, pg, rank, world, args.steps)
    if pg is not None:
        check_rank_digests(job, pg, rank, world, hb, 'after the timed steps')
    res = None
    if rank == 0:
        res = {'metric': 'dry-run (no GPU work; launcher / rendezvous / N-rank skeleton check only)',
               'value': None, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none',
               'data': 'none',
               'config': {'workload': 'dry-run', 'parallelism': 'dp%d' % world,
                          'rccl_world_size': world,
                          'pipelined_update': job.route()['pipelined'],
                          'sharded_update': job.route()['sharded'],
                          'ms_per_step_rank_min': round(min(rank_dt) / args.steps * 1e3, 3),
                          'ms_per_step_rank_max': round(max(rank_dt) / args.steps * 1e3, 3),
                          'launched_by_bench': os.environ.get('NAWS_BENCH_CHILD') == '1'}}
        res.update(extra)
        res.update(fallback_keys())
        res['config'].update({k: v for k, v in res.items()
                              if k in extra or k.startswith('route_') or k == 'supervised'})
        save_main_line(res)
    hb.phase('main_done')
    if pg is not None and not args.no_route_ab:
        legs = route_ab(job, pg, rank, world, args.steps, hb, 1)
        if rank == 0:
            res.update(legs)
            res['config'].update(legs)
    hb.phase('emit')
    if pg is not None:
        dist.destroy_process_group()
    if rank == 0:
        print('rank 0: some library banner after which the line must still come last')
        emit(res)
    else:
        print('rank %d: noise on stdout' % rank, flush=True)
    hb.phase('done', printed=True)


def main():
    args = parse()
    if args.infer:
        return infer_main(args)
    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if not under_launcher and (args.gpus > 1 or args.self_launch):
        # a bare `python bench.py --gpus N`: this process becomes the launcher (before anything
        # here has touched the GPU) and the ranks run as its children
        return self_launch(args, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        sys.exit('bench.py: --gpus %d inside a %d-rank job (WORLD_SIZE=%d): launch it with '
                 '--nproc-per-node %d, or run `python bench.py --gpus %d` bare and it starts its '
                 'own ranks' % (args.gpus, world, world, args.gpus, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.drop_sharded_update:
        args.sharded_update = False
    if world > 1 and not args.no_supervisor and os.environ.get('NAWS_BENCH_WORKER') != '1':
        # A rank of an N > 1 job: THIS process stays a supervisor (it has not touched the GPU and
        # never will) and runs the rank as a watched child - per-phase watchdog; after a stall, a
        # death or a rank-digest mismatch every rank's supervisor starts a FRESH worker on the
        # next route of the ladder (naws_hip/supervise.py)
        from naws_hip.supervise import supervise_rank
        sys.exit(supervise_rank(os.path.abspath(__file__), sys.argv[1:]))
    if args.dry_run:
        return dry_run(args, rank, world)
    from naws_hip.supervise import Heartbeat
    hb = Heartbeat()
    hb.phase('init')
    import torch
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pg = None
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if args.share_gpu:
            pg = init_process_group('gloo', rank, world)
        else:
            pg = init_process_group('nccl', rank, world, dev)
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine

    num_fg = args.classes
    B = args.images_per_gpu
    hb.phase('setup')
    eng = WsddnEngine(num_fg + 1, dev, dilation=2, dropout=0.5, is_mean=True, momentum=0.9,
                      weight_decay=5e-4, iter_size=1, gpu_num=world * B, seed=11,
                      process_group=pg, world_size=world, allreduce_chunks=args.allreduce_chunks,
                      mfma_dtype=args.mfma_dtype, sharded_update=args.sharded_update, rank=rank,
                      pipeline_update=False if args.no_pipeline_update else None)
    if args.force_dist:
        eng.reducer.force = True
    if args.no_conv_streams:
        eng.conv_streams = False
    # Rank 0 holds the seeded initial weights; every other rank starts from a DIFFERENT seed and
    # receives rank 0's blobs by broadcast (reference: utils/net_wsl.py:183-207 copies GPU 0's
    # blobs to the other GPUs) - identical seeds on every rank would hide a broken broadcast
    blobs = synthetic.init_blobs(num_fg, seed=11 if rank == 0 else 1000 + rank)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    del blobs
    roidb = synthetic.make_roidb(B, args.rois, num_fg, args.height, args.width, seed=11 + rank)
    mb = synthetic.make_minibatch(roidb, num_fg)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    import numpy as np
    counts = np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)
    seg = [0] + np.cumsum(counts).tolist()        # per-image row offsets, known on the host
    eng.set_lr(args.lr)
    job = EngineJob(eng, t, seg, two_call=args.no_fused_update)
    nrank = {}
    if pg is not None:
        nrank.update(n_rank_prologue(job, pg, rank, world, args, hb))

    # live timing of the dominant kernel (fc6 forward GEMM, N = 8192) with HIP events on
    # the launch stream
    ev = []
    pev = []
    uev = []     # the fused SGD kernel, on the update stream
    cev = []     # update stream: from "gradients complete" to "all-reduce complete"

    job.timed_hooks = (ev, pev, uev, cev)
    inject = os.environ.get('NAWS_BENCH_INJECT', '').split(':')   # tests: 'stall:<route>:<rank>'

    def injected_stall():
        if len(inject) >= 3 and inject[0] == 'stall' and int(inject[2]) == rank:
            r = job.route()
            name = 'sharded' if r['sharded'] else 'pipelined' if r['pipelined'] else 'unpipelined'
            if inject[1] in (name, 'any'):
                time.sleep(3600)           # (host-side: the other ranks wait in their collectives)

    hb.phase('warmup', steps=args.warmup)
    job.probe_fn = step0_probe          # the in-run parity check compares every plan's FIRST step
    for i in range(args.warmup):
        job.step(False)
    eng.flush()
    if pg is not None:
        # every rank must hold bit-identical parameters, momentum, operand planes and scales
        # after W real updates - BEFORE anything is timed (exit 4 otherwise)
        nrank['rank_digest_equal'] = check_rank_digests(job, pg, rank, world, hb, 'after warm-up')
    hb.phase('timed', steps=args.steps)
    injected_stall()
    eng.reducer.log = msg_log = []
    dt, rank_dt = timed_steps(job, pg, rank, world, args.steps)
    eng.reducer.log = None
    out = job.last
    probe0 = [job.first_probe]
    if pg is not None:
        check_rank_digests(job, pg, rank, world, hb, 'after the timed steps')
    hb.phase('extras')
    loss = float(out['loss_cls'].sum().item() + out['loss_cls_noise'].sum().item())

    # after the timed steps: the conv body alone (nothing else on the device).  Inside a step it
    # shares the chip with the previous step's parameter update (side stream), so its stage time
    # there reads longer than the stack itself takes.
    # One rank, fc6_w updated inside its wgrad GEMM: also time the route with a gradient between
    # wgrad and update - the one every rank of an N > 1 job takes - so that a scaling series can be
    # read against a like-for-like N = 1 number (reported beside `value`, never as `value`).
    deferred_ms = None
    if world == 1 and not args.no_fused_update and eng._can_fuse_wgrad_update():
        eng.timing_events = eng.phase_events = eng.update_events = eng.comm_events = None
        n_def = args.steps               # the same number of steps as `value`
        for it in range(5 + n_def):
            if it == 5:
                eng.flush()
                torch.cuda.synchronize()
                d0 = time.perf_counter()
            eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        deferred_ms = (time.perf_counter() - d0) / n_def * 1e3
    # One rank: what the N-rank schedule costs THIS GPU when something occupies compute units and
    # HBM the way the RCCL exchange would (reducer.EmulatedExchange; a projection, labelled so)
    projections = {}
    if world == 1 and args.mfma_dtype == 'fp16x2' and not args.share_gpu and not args.force_dist \
            and not args.no_projection:
        spec = [x for x in args.emulate_exchange.split(',') if x]
        # Two paces per rank count: the links delivering 0.6 of their
        # 153 GB/s per direction (the optimistic point of rounds 4-5) and 0.3 of it (RCCL's bus
        # bandwidth on ~1 GB messages is commonly well below the link rate) - key suffix _pace03
        cases = [(int(spec[0]), int(spec[1]) if len(spec) > 1 else 32,
                  float(spec[2]) if len(spec) > 2 else None, '')] if spec else \
            [(n, 32, frac * LINK_GBPS * min(n - 1, 7), tag)
             for frac, tag in ((0.6, ''), (0.3, '_pace03')) for n in (2, 4, 8)]
        psteps = max(20, args.steps // 3)
        for n, cus, gbps, tag in cases:
            projections['%d%s' % (n, tag)] = project_n_ranks(eng, t, seg, n, cus, gbps, psteps)
            # the same schedule without the piece-by-piece update (round 4's route), for the A/B
            projections['%d_unpipelined%s' % (n, tag)] = project_n_ranks(
                eng, t, seg, n, cus, gbps, psteps, pipeline=False)
        for n, cus, gbps, tag in cases:
            if n == cases[-1][0]:
                projections['%d_sharded%s' % (n, tag)] = project_n_ranks(eng, t, seg, n, cus, gbps,
                                                                        psteps, sharded=True)
    conv_alone_ms = None
    roipool_alone_ms = None
    if rank == 0:
        eng.timing_events = eng.phase_events = eng.update_events = eng.comm_events = None
        eng.conv_body(t['data'])
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            eng.conv_body(t['data'])
        c1.record()
        torch.cuda.synchronize()
        conv_alone_ms = c0.elapsed_time(c1) / 10
        # RoIPoolF + boost alone (one launch over all proposals, maps given): inside a step each
        # image's proposals are pooled at the tail of that image's conv chain, so the step has no
        # RoIPool stage of its own to time
        if args.mfma_dtype == 'fp16x2' and eng._amax5 is not None:
            from naws_hip import ops as _ops
            conv5 = eng.conv_body(t['data'])
            maps = eng._roi_maps
            torch.cuda.synchronize()
            if maps is not None:
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(10):
                    _ops.roi_pool_f_f16x2(conv5, t['rois'], eng._amax5, eng.roi_size, eng.roi_size,
                                          eng.spatial_scale, boost=t['obn_scores'].reshape(-1),
                                          hier=True, maps=maps)
                c1.record()
                torch.cuda.synchronize()
                roipool_alone_ms = c0.elapsed_time(c1) / 10
            del conv5, maps

    if rank == 0:
        rt = mb['rois'].shape[0]
        k6 = 512 * 49
        fc6_flops = 2.0 * rt * (2 * 4096) * k6
        kern_ms = sum(s.elapsed_time(e) for s, e in ev) / max(args.steps, 1)    # (N > 1: two pieces per step)
        achieved = fc6_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
        # per-stage wall time on the main stream (HIP events), averaged over the timed steps
        stages, order = {}, []
        for (n0, e0), (n1, e1) in zip(pev[:-1], pev[1:]):
            if n1 == 'start':
                continue
            if n1 not in stages:
                stages[n1] = []
                order.append(n1)
            stages[n1].append(e0.elapsed_time(e1))
        stage_ms = {n: round(sum(v) / len(v), 3) for n, v in stages.items()}
        bf = args.mfma_dtype == 'bf16'
        x3 = args.mfma_dtype == 'fp32x3'
        h2 = args.mfma_dtype == 'fp16x2'
        peak, kname = plan_peak_and_kernel(args.mfma_dtype)
        dtype = ('bf16 (bf16 MFMA operands, fp32 accumulate/storage/loss)' if bf else
                 'f32 (exact 3xbf16 operand split, 6-pass bf16 MFMA, fp32 accumulate)' if x3 else
                 'f32 (2xf16 split, 3-pass f16 MFMA, fp32 accumulate)' if h2 else
                 'f32 (fp32 MFMA)')
        arith = ('bf16 MFMA conv/fc6/fc7, fp32 storage/fc8/loss/SGD' if bf else
                 'fp32 via exact 3xbf16 splits (fc6/fc7, conv1_2-2_2), rest fp32 MFMA' if x3 else
                 'fp32 via 2xf16 row-scaled splits (fc6/fc7, conv1_2-5_3), rest fp32' if h2 else
                 'fp32 MFMA')
        # algorithmic work of the configuration's stages (per image at 600x1000,
        # R = 2000: conv 463.7 GFLOP; RoIPool+boost reads the 18.8 MB feature map and writes the
        # 200.7 MB roi_feat; SGD moves 5 x 957.7 MB)
        headline = (args.height, args.width, args.rois, num_fg) == (600, 1000, 2000, 20)
        conv_gflop = 463.7 * B if headline else None
        roipool_bytes = B * (512 * 74 * 124 * 4 + args.rois * 25088 * 4 + args.rois * 24) if headline else None
        # 3 reads + 2 writes of the arena; the fp16x2 plan's update also writes the hi / lo operand
        # planes of fc6_w / fc7_w (4 B per weight) in the same kernel
        sgd_bytes = 5 * 4 * eng.arena.total
        plane_bytes = 4 if h2 else 6 if x3 else 2 if bf else 0      # per fc6 / fc7 weight
        if plane_bytes and eng.fused_planes and eng._sgd_regions is not None:
            sgd_bytes += plane_bytes * 2 * 4096 * (eng.k6 + 4096)
        wgrad_update = (not args.no_fused_update) and eng._can_fuse_wgrad_update()
        if wgrad_update:       # the deferred kernel no longer touches fc6_w (nor reads its gradient)
            sgd_bytes -= (5 * 4 + plane_bytes) * 2 * 4096 * eng.k6
        sgd_ms = sum(s.elapsed_time(e) for s, e in uev) / max(len(uev), 1)
        roof = {'bound': 'mfma', 'kernel': 'fc6 fwd (both branches, M=%d N=8192 K=%d): %s' % (rt, k6, kname),
                'achieved': round(achieved, 2) if achieved else None,
                'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4) if achieved else None,
                'traffic': None, 'kernel_ms': round(kern_ms, 4)}
        if (x3 or h2) and achieved:
            # 6 (3) executed 16-bit MFMA flops per algorithmic flop: the same fraction against
            # the instruction's own peak
            npass = 6 if x3 else 3
            roof.update(executed_tflops=round(npass * achieved, 1),
                        executed_peak=BF16_MFMA_PEAK_TFLOPS, mfma_passes=npass)
        # the conv stack against the fp32 MFMA peak, RoIPool and
        # SGD against the achievable HBM rate - all from HIP events of THIS run (the conv body and
        # RoIPool stages on the main stream; the SGD kernel on the update stream, where it runs
        # underneath the next iteration's conv body)
        if conv_gflop and stage_ms.get('conv_body'):
            tf = conv_gflop / stage_ms['conv_body']
            roof.update(conv_stack_ms=stage_ms['conv_body'], conv_stack_tflops=round(tf, 1))
            if conv_alone_ms:
                roof.update(conv_stack_alone_ms=round(conv_alone_ms, 3),
                            conv_stack_alone_tflops=round(conv_gflop / conv_alone_ms, 1))
            # THE fraction: what the stack EXECUTES on its matrix pipe against that pipe's peak.
            # Split plans: npass passes over the direct layers (conv1_2..conv3_3: 221.0 of the
            # 463.7 algorithmic GFLOP per image) and over Winograd's share of conv4_1..conv5_3
            # (242.7 GFLOP: 1/2.25 of it under F(2x2,3x3), 1/4 under F(4x4,3x3)); conv1_1 runs on
            # the vector unit.
            if h2 or x3:
                npass = 3 if h2 else 6
                # fp16x2: F(4x4,3x3) from WINO_F4_MIN_CIN input channels (conv4_2..conv5_3, 220.6 of
                # the deep layers' 242.7 GFLOP; executed with the tile padding: 19 x 32 tiles of
                # 4 x 4 for 75 x 125 outputs = 1.038 x), F(2x2) for the rest of them
                f4 = (220.6 if 0 < getattr(eng, 'WINO_F4_MIN_CIN', 0) <= 512 else 0.0) if h2 else 0.0
                if h2 and 0 < getattr(eng, 'WINO_F4_MIN_CIN', 0) <= 256:
                    f4 = 242.7
                ex_gflop = conv_gflop / 463.7 * npass * (221.0 + (242.7 - f4) / 2.25 + 1.038 * f4 / 4.0)
                roof['conv_stack_winograd_f4_gflop_of_463_7'] = f4
                tag = 'f16' if h2 else 'bf16'
                roof['conv_stack_executed_%s_tflops' % tag] = round(ex_gflop / stage_ms['conv_body'], 1)
                roof['conv_stack_frac_vs_%s_mfma_peak' % tag] = round(
                    ex_gflop / stage_ms['conv_body'] / BF16_MFMA_PEAK_TFLOPS, 3)
                if conv_alone_ms:
                    roof['conv_stack_alone_frac_vs_%s_mfma_peak' % tag] = round(
                        ex_gflop / conv_alone_ms / BF16_MFMA_PEAK_TFLOPS, 3)
                # NOT a fraction: how many times faster than the fp32-MFMA floor (all 463.7
                # algorithmic GFLOP per image at 157.3 TFLOP/s) - above 1 because the work is not
                # on the fp32 pipe, not because any of it is skipped
                roof['conv_stack_speedup_over_fp32_mfma_floor'] = round(tf / FP32_MFMA_PEAK_TFLOPS, 3)
            elif bf:      # one pass, every layer direct (no Winograd in this plan)
                roof['conv_stack_frac_vs_bf16_mfma_peak'] = round(tf / BF16_MFMA_PEAK_TFLOPS, 3)
            else:
                roof['conv_stack_frac_vs_fp32_mfma_peak'] = round(tf / FP32_MFMA_PEAK_TFLOPS, 3)
        if roipool_bytes and roipool_alone_ms:
            # (pooled per image on the conv streams inside a step: `stage_ms_roi_pool` is what is
            # left on the main stream - the join -, `stage_ms_conv_body` includes the pooling)
            gbs = roipool_bytes / roipool_alone_ms / 1e6
            roof.update(roipool_ms=round(roipool_alone_ms, 4), roipool_GBps=round(gbs, 1),
                        roipool_frac_vs_hbm_6300=round(gbs / HBM_ACHIEVABLE_GBPS, 3),
                        roipool_measured='alone, one launch over all proposals, after the timed steps')
        elif roipool_bytes and stage_ms.get('roi_pool'):
            gbs = roipool_bytes / stage_ms['roi_pool'] / 1e6
            roof.update(roipool_ms=stage_ms['roi_pool'], roipool_GBps=round(gbs, 1),
                        roipool_frac_vs_hbm_6300=round(gbs / HBM_ACHIEVABLE_GBPS, 3))
        if sgd_ms > 0:
            gbs = sgd_bytes / sgd_ms / 1e6
            roof.update(sgd_ms=round(sgd_ms, 4), sgd_GBps=round(gbs, 1),
                        sgd_frac_vs_hbm_6300=round(gbs / HBM_ACHIEVABLE_GBPS, 3))
        # HBM traffic of the dominant kernel is NOT measured by this run: it comes from the
        # rocprofv3 PMC passes of this same command (profiles/rNN_bench_*traffic.json, written by
        # tools/summarize_profile.py); the field says where it was read from
        import glob
        tjd, tsrc = committed_traffic(args.mfma_dtype)
        roof['traffic_measured_in_run'] = False
        if tjd is not None and headline and B == 2:
            roof['traffic'] = tjd['hbm_bytes_per_launch']
            roof['traffic_source'] = tsrc
            roof['traffic_ratio_vs_algorithmic'] = round(
                tjd['hbm_bytes_per_launch'] / tjd['algorithmic_bytes_per_launch'], 3)
            # the same ratio (counter bytes / algorithmic bytes per launch) for the other hot
            # kernels of the plan, from the same PMC passes (tools/summarize_profile.py EXTRA)
            for e in tjd.get('other_kernels', []):
                key = e['kernel'].split(' (')[0].replace(' ', '_').replace('<', '_').replace('>', '') \
                    .replace(',', '_').replace('+', 'and').replace('/', '_')
                roof['traffic_ratio_' + key] = e['ratio']
        pmc = {'fp16x2': sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_default_plan_pmc.md'))),
               'fp32x3': [os.path.join(ROOT, 'profiles', 'r01_x3_gemm_pmc.md')]}
        if pmc.get(args.mfma_dtype):
            roof['profile_ref'] = os.path.relpath(pmc[args.mfma_dtype][-1], ROOT)
        cfg = {'workload': 'configs[1] flickr_voc na_wsddn_V-16-C5_1x C=%d: %d img %dx%d/GPU x %d '
                           'rois, fwd+bwd+allreduce+SGD' % (num_fg, B, args.height, args.width,
                                                            args.rois),
               'arithmetic': arith,
               'global_batch_images': world * B, 'parallelism': 'dp%d' % world,
               'lr': args.lr, 'final_loss': round(loss, 5),
               'rccl_backend': (torch.distributed.get_backend() if pg is not None else 'none'),
               'rccl_world_size': (torch.distributed.get_world_size() if pg is not None else 1),
               'allreduce_chunks': eng.allreduce_chunks if eng.reducer.active else 0,
               'shared_gpu': bool(args.share_gpu),
               'sharded_update': bool(eng._shard_blocks() is not None),
               'pipelined_update': bool(eng._pipelined()),
               # where fc6_w (86 % of the parameters) is updated: 'wgrad_epilogue' = inside its
               # weight-gradient GEMM, possible only without a gradient exchange (one rank);
               # 'deferred_kernel' = gradient written, (all-reduced,) then the SGD kernel on the
               # side stream - what every rank does when world_size > 1
               'fc6_update_path': 'wgrad_epilogue' if wgrad_update else 'deferred_kernel',
               'deferred_route_ms_per_step': None if deferred_ms is None else round(deferred_ms, 3),
               'deferred_route_images_per_sec': (None if deferred_ms is None
                                                 else round(world * B / deferred_ms * 1e3, 3)),
               # per-rank wall time of the timed region (value uses the max); the time the main
               # stream stood waiting for the deferred all-reduce + SGD + weight re-split before
               # the head could read the parameters (HIP events around engine.flush(): the
               # exposed part of the exchange and update - everything else was hidden under the
               # conv body); the time the update stream waited for the collective after this
               # rank's gradients were complete
               'ms_per_step_rank_min': round(min(rank_dt) / args.steps * 1e3, 3),
               'ms_per_step_rank_max': round(max(rank_dt) / args.steps * 1e3, 3),
               'exposed_comm_ms': None,
               'allreduce_wait_ms': (round(sum(s.elapsed_time(e) for s, e in cev) / len(cev), 3)
                                     if cev else None)}
        # the messages THIS rank handed to the exchange in one step, in order ("kind:elements"),
        # and whether that is the schedule the N-rank projection of the one-GPU line plays
        # (project_n_ranks: 4 fc6_w row chunks at 2 ranks, 2 above, then the small gradients)
        if msg_log:
            per = msg_log[:len(msg_log) // args.steps]
            cfg['exchange_messages_per_step'] = ','.join('%s:%d' % (k, n) for k, n in per)
            cfg['exchange_bytes_per_step'] = 4 * sum(n for _k, n in per)
            if not args.sharded_update:
                from naws_hip.reducer import message_plan, message_slice
                plan = message_plan(eng.arena, 8192, 4 if world == 2 else 2, True, eng._pipelined())
                want = [['all_reduce', int(message_slice(eng.arena, eng.grads, k, r, eng.k6).numel())]
                        for k, r in plan]
                cfg['exchange_schedule_equals_projection'] = bool(
                    [list(m) for m in per] == want) if world > 1 else None
        # which route the two-pace projection favours at every (rank count, pace): the route that
        # never loses by more than 0.3 ms anywhere is the one an N > 1 job should default to
        verdict, worst = [], {'pipelined': 0.0, 'unpipelined': 0.0}
        for key in sorted(k for k in projections if '_' not in str(k).replace('_pace03', '')):
            n_, tag = str(key).replace('_pace03', ''), ('_pace03' if 'pace03' in str(key) else '')
            a = projections[key]['ms_per_step']
            b = projections.get('%s_unpipelined%s' % (n_, tag), {}).get('ms_per_step')
            if b is None:
                continue
            verdict.append('n%s%s %+.2f' % (n_, tag, a - b))
            worst['pipelined'] = max(worst['pipelined'], a - b)
            worst['unpipelined'] = max(worst['unpipelined'], b - a)
        if verdict:
            cfg['projected_pipelined_minus_unpipelined_ms'] = ' '.join(verdict)
            cfg['projected_worst_loss_ms_pipelined'] = round(worst['pipelined'], 3)
            cfg['projected_worst_loss_ms_unpipelined'] = round(worst['unpipelined'], 3)
        for n, pr in sorted(projections.items(), key=lambda kv: str(kv[0])):
            # PROJECTIONS, not measurements of an N-GPU job: the one-rank step with the N-rank
            # schedule and a paced copy kernel in the all-reduce's place (see EmulatedExchange);
            # "<N>_sharded" = the NAWS.SHARDED_UPDATE schedule (reduce-scatter, 1 / N of fc6_w
            # updated here, all-gather)
            nn = int(str(n).split('_')[0])
            cfg['projected_ms_per_step_n%s' % n] = pr['ms_per_step']
            cfg['projected_images_per_sec_n%s' % n] = round(nn * B / pr['ms_per_step'] * 1e3, 1)
            cfg['projected_stages_n%s' % n] = ' '.join('%s %.2f' % kv for kv in pr['stages'].items()) + \
                ' | between steps %.2f' % pr['between_steps_ms']
            cfg['projected_messages_n%s' % n] = ','.join('%s:%d' % (k, m) for k, m in pr['messages_per_step'])
            cfg['projected_exchange_n%s' % n] = (
                '%d CUs x %.0f GB/s, %.2f GB per step, allreduce_chunks %d, %s, join %.2f ms, head '
                'forward %.2f ms (%.2f without an exchange)'
                % (pr['cus'], pr['gbps'], pr['bytes_per_step'] / 1e9, pr['chunks'],
                   'update piece by piece' if pr['pipelined'] else 'one update launch',
                   pr['exposed_ms'], pr['head_fwd_ms'], stage_ms.get('head_fwd', float('nan'))))
        for k, v in stage_ms.items():                 # flat: the driver's parser drops nested dicts
            cfg['stage_ms_' + k] = v
        cfg['exposed_comm_ms'] = stage_ms.get('join_update')
        res = {
            'metric': 'images/sec (600px, 2000 proposals) VGG16-C5 WSDDN fwd+bwd',
            'value': round(world * B * args.steps / dt, 3),
            'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic', 'config': cfg, 'roofline': roof,
        }
        # A scaling series must compare like with like: every rank of an N > 1 job writes
        # fc6_w's gradient, exchanges it and updates in the deferred kernel, while `value` at
        # N = 1 updates fc6_w inside its wgrad GEMM.  This field is the N = 1 point of the N > 1
        # ALGORITHM (same steps, same clock as `value`); at N > 1 it equals `value`.
        res['scaling_baseline_value'] = (round(world * B / deferred_ms * 1e3, 3)
                                         if deferred_ms is not None else res['value'])
        if world == 1 and args.mfma_dtype == 'fp16x2' and not args.no_alt_plan:
            # the same workload, measured in this run, (a) with the exact 3 x bf16 operand split
            # and (b) with every GEMM on v_mfma_f32_32x32x2_f32, for readers who want the number
            # without the 2 x f16 operand representation
            del eng
            torch.cuda.empty_cache()
            for mode, key in (('fp32x3', 'fp32x3_plan'), ('fp32', 'fp32_mfma_plan')):
                res[key] = alt_plan(args, dev, num_fg, B, t, seg, mode)
                cfg[key + '_images_per_sec'] = res[key]['value']
                cfg[key + '_ms_per_step'] = res[key]['ms_per_step']
                # flat copies of the plan's own roofline block (nested dicts may be dropped)
                roof[key + '_fc6_fwd_kernel_ms'] = res[key]['roofline']['kernel_ms']
                roof[key + '_fc6_fwd_tflops'] = res[key]['roofline']['achieved']
                roof[key + '_fc6_fwd_peak'] = res[key]['roofline']['peak']
                roof[key + '_fc6_fwd_frac'] = res[key]['roofline']['frac']
                roof[key + '_conv_stack_ms'] = res[key]['roofline'].get('conv_stack_ms')
            if not args.no_parity_check:
                res['parity_in_run'] = parity_in_run(args, dev, num_fg, B, t, seg, probe0[0], res)
                for k, v in res['parity_in_run'].items():
                    cfg['parity_' + k] = v
            for key in ('fp32x3_plan', 'fp32_mfma_plan'):
                res[key].pop('probe', None)
            if not args.no_extra_configs and headline and B == 2:
                extra_configs(args, dev, B, res, cfg, roof)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, num_fg)
        # N > 1 (or --force-dist): the broadcast / bare-exchange / rank-digest keys, flat in the
        # line and in config (the driver's parser keeps flat keys), and what the supervisor did
        res.update(nrank)
        res.update(fallback_keys())
        cfg.update(nrank)
        cfg.update(fallback_keys())
        save_main_line(res)
    hb.phase('main_done')
    if pg is not None and not args.no_route_ab and args.mfma_dtype == 'fp16x2':
        # the other update routes, K steps each, in this same job (value stays the default route's)
        job.timed_hooks = None
        legs = route_ab(job, pg, rank, world, args.steps, hb, B)
        if rank == 0:
            res.update(legs)
            res['config'].update(legs)
    hb.phase('emit')
    if pg is not None:
        torch.distributed.destroy_process_group()
    if rank == 0:
        emit(res)
        if res.get('parity_in_run', {}).get('ok') is False:
            sys.exit('bench.py: the headline plan disagrees with the fp32 plans beyond the stated '
                     'tolerances: %s' % json.dumps(res['parity_in_run']))
    hb.phase('done', printed=True)


if __name__ == '__main__':
    main()
