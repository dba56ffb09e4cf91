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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, dense v_mfma_f32_32x32x16_bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-plan', action='store_true',
                    help='skip the extra fp32-MFMA-only measurement of the same workload')
    ap.add_argument('--allreduce-chunks', type=int, default=0, help='0 = auto (engine.py)')
    ap.add_argument('--no-conv-x3', action='store_true',
                    help='fp32x3 plan: keep the conv body on the fp32 MFMA (direct + Winograd)')
    ap.add_argument('--wino-x3', action='store_true',
                    help='fp32x3 plan: run the Winograd batched GEMMs in the split too (no gain)')
    ap.add_argument('--no-fuse-pool', action='store_true',
                    help='fp16x2 plan: pool1..pool3 as separate kernels instead of in the conv epilogue')
    ap.add_argument('--no-roi-planes', action='store_true',
                    help='fp16x2 plan: RoIPoolF writes fp32 features that are then split (two '
                         'more passes) instead of writing the fc6 operand planes itself')
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
    return ap.parse_args()


def cpu_baseline(args, num_fg):
    """CPU restatement of the reference path (oracle/, torch-CPU + C ops) on BASELINE
    configs[0]: 2 synthetic 600x1000 images x 500 proposals, one fwd+bwd+SGD iteration."""
    import numpy as np
    import torch
    from detectron.datasets import synthetic
    from oracle import oracle
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    roidb = synthetic.make_roidb(2, args.cpu_rois, num_fg, args.height, args.width, seed=11)
    mb = synthetic.make_minibatch(roidb, num_fg)
    blobs = synthetic.init_blobs(num_fg, seed=11)
    rt = mb['rois'].shape[0]
    rng = np.random.default_rng(0)
    masks = {k: (rng.uniform(size=(rt, 4096)) > 0.5).astype(np.float32)
             for k in ('drop6', 'drop7', '_[noisy]_drop6', '_[noisy]_drop7')}
    t0 = time.time()
    ref = oracle.full_forward_backward(blobs, mb, masks, num_fg)
    lr = np.array([1e-3], np.float32)
    for name, g in ref['grads'].items():
        p = blobs[name].numpy().reshape(-1)
        m = np.zeros_like(p)
        a = np.zeros_like(p)
        bias = name.endswith('_b')
        oracle.acm_sgd(np.ascontiguousarray(g.reshape(-1)), m, lr, p, a, 0.9, 0,
                       0.0 if bias else 5e-4, 1, 2, 2.0 if bias else 1.0, 0)
    dt = time.time() - t0
    return {'value': round(2.0 / dt, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '1 iteration (fwd+bwd+SGD) of 2 images %dx%d x %d proposals, fp32, '
                      'torch-CPU conv/fc + C oracle ops, %.1f s' % (args.height, args.width,
                                                                    args.cpu_rois, dt)}


def alt_plan(args, dev, num_fg, B, t, seg, mode):
    import torch
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    eng = WsddnEngine(num_fg + 1, dev, dilation=2, dropout=0.5, is_mean=True, momentum=0.9,
                      weight_decay=5e-4, iter_size=1, gpu_num=B, seed=11, mfma_dtype=mode)
    blobs = synthetic.init_blobs(num_fg, seed=11)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    del blobs
    eng.set_lr(args.lr)
    steps = max(1, min(args.steps, 5))

    def run(n):
        for _ in range(n):
            out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        return out
    run(2)
    t0 = time.perf_counter()
    out = run(steps)
    dt = time.perf_counter() - t0
    return {'mfma_dtype': mode, 'value': round(B * steps / dt, 3), 'unit': 'images/sec',
            'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'warmup': 2,
            'final_loss': round(float(out['loss_cls'].sum().item() +
                                      out['loss_cls_noise'].sum().item()), 5)}


def main():
    args = parse()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pg = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine

    num_fg = args.classes
    B = args.images_per_gpu
    eng = WsddnEngine(num_fg + 1, dev, dilation=2, dropout=0.5, is_mean=True, momentum=0.9,
                      weight_decay=5e-4, iter_size=1, gpu_num=world * B, seed=11,
                      process_group=pg, world_size=world, allreduce_chunks=args.allreduce_chunks,
                      mfma_dtype=args.mfma_dtype)
    if args.force_dist:
        eng.reducer.force = True
    if args.no_conv_x3:
        eng.conv_x3 = False
    if args.wino_x3:
        eng.wino_x3 = True
    if args.no_conv_streams:
        eng.conv_streams = False
    if args.no_roi_planes:
        eng.roi_planes = False
    if args.no_fuse_pool:
        eng.fuse_pool = False
    blobs = synthetic.init_blobs(num_fg, seed=11)     # identical on every rank (= broadcast)
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

    # live timing of the dominant kernel (fc6 forward GEMM, N = 8192) with HIP events on
    # the launch stream
    ev = []

    pev = []

    def step(timed):
        if timed:
            eng.timing_events = ev
            eng.phase_events = pev
        else:
            eng.timing_events = None
            eng.phase_events = None
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
        eng.sgd_step()
        return out

    for _ in range(args.warmup):
        out = step(False)
    eng.flush()
    if pg is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    eng.flush()          # the last iteration's (deferred) all-reduce + SGD belongs to the K steps
    if pg is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        td = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(td, op=torch.distributed.ReduceOp.MAX)
        dt = float(td.item())
    loss = float(out['loss_cls'].sum().item() + out['loss_cls_noise'].sum().item())

    if rank == 0:
        rt = mb['rois'].shape[0]
        k6 = 512 * 49
        fc6_flops = 2.0 * rt * (2 * 4096) * k6
        kern_ms = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
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
        # fp32x3 executes 6 bf16 MFMA flops per algorithmic fp32 flop: its ceiling in algorithmic
        # TFLOP/s is the bf16 dense peak / 6 (frac = executed MFMA flops / bf16 peak)
        peak = (BF16_MFMA_PEAK_TFLOPS if bf else
                round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1) if x3 else
                round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1) if h2 else FP32_MFMA_PEAK_TFLOPS)
        kname = ('gemm_x3_kernel<256,256,2x4 waves,2 stages,1 plane x 4 K-slabs> (bf16 slab operands)' if bf else
                 'gemm_x3_kernel<256,256,2x4 waves,3 stages> = 6 x v_mfma_f32_32x32x16_bf16 per '
                 'fp32 product' if x3 else
                 'gemm_x3_kernel<256,256,2x4 waves,2 stages,2 planes x 2 K-slabs,f16> = 3 x '
                 'v_mfma_f32_32x32x16_f16 per fp32 product' if h2 else
                 'gemm_f32_kernel<256,256,16,KC,KC,4x4 waves>')
        res = {
            'metric': 'images/sec (600px, 2000 proposals) VGG16-C5 WSDDN fwd+bwd',
            'value': round(world * B * args.steps / dt, 3),
            'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if bf else 'f32', 'data': 'synthetic',
            'config': {'workload': 'flickr_voc na_wsddn_V-16-C5_1x (= BASELINE.json webly_wsddn_V-16-C5_1x; C=%d): %d images %dx%d per '
                                   'GPU x %d proposals, fwd+bwd+allreduce+SGD, %s' % (
                                       num_fg, B, args.height, args.width, args.rois,
                                       'bf16 MFMA conv/fc6/fc7, fp32 storage/fc8/loss/SGD'
                                       if bf else 'fp32; fc6/fc7 GEMMs as exact 3xbf16 splits on '
                                       'the bf16 MFMA (fp32-accurate), conv/fc8 on the fp32 MFMA'
                                       if x3 else 'fp32; fc6/fc7 GEMMs and conv1_2..conv5_3 as power-of-two-'
                                       'scaled 2xf16 operand splits on the f16 MFMA (fp32 '
                                       'accumulate, operand error 2^-22); conv1_1, fc8, '
                                       'softmaxes, loss, SGD in fp32'
                                       if h2 else 'fp32 MFMA'),
                       'global_batch_images': world * B, 'parallelism': 'dp%d' % world,
                       'lr': args.lr, 'final_loss': round(loss, 5), 'stage_ms': stage_ms},
            'roofline': {'bound': 'mfma', 'kernel': '%s (fc6 fwd, both branches, M=%d N=8192 K=%d)' % (
                kname, rt, k6),
                         'achieved': round(achieved, 2) if achieved else None,
                         'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4) if achieved else None,
                         'traffic': None, 'kernel_ms': round(kern_ms, 4)},
        }
        if (x3 or h2) and achieved:
            # 6 (3) executed 16-bit MFMA flops per algorithmic flop: the same fraction against
            # the instruction's own peak
            npass = 6 if x3 else 3
            res['roofline'].update(executed_tflops=round(npass * achieved, 1),
                                   executed_peak=BF16_MFMA_PEAK_TFLOPS,
                                   note='achieved/peak are algorithmic fp32 TFLOP/s against '
                                        '16-bit dense MFMA peak / %d passes; PMC (profiles/%s): '
                                        'MFMA pipe busy ~70%% of SIMD-cycles at a power-throttled '
                                        '%s GHz, 0 LDS bank conflicts' % (
                                            npass, 'r01_x3_gemm_pmc.md' if x3 else 'r01_h2_gemm_pmc.md',
                                            '1.84' if x3 else '1.5'))
        # HBM traffic of that kernel comes from the rocprofv3 PMC passes of this same command
        # (profiles/rNN_bench_traffic.json, written by tools/summarize_profile.py)
        import glob
        tj = [f for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench*traffic*.json')))
              if json.load(open(f)).get('mfma_dtype', 'fp32') == args.mfma_dtype]
        if tj and args.rois == 2000 and B == 2:
            res['roofline']['traffic'] = json.load(open(tj[-1]))['hbm_bytes_per_launch']
            res['roofline']['traffic_source'] = os.path.relpath(tj[-1], ROOT)
        if world == 1 and args.mfma_dtype in ('fp16x2', 'fp32x3') and not args.no_alt_plan:
            # the same workload with every GEMM on v_mfma_f32_32x32x2_f32, measured in this run,
            # for readers who want the number without the 3xbf16 operand split
            del eng
            torch.cuda.empty_cache()
            res['fp32_mfma_plan'] = alt_plan(args, dev, num_fg, B, t, seg, 'fp32')
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, num_fg)
        print(json.dumps(res))
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
