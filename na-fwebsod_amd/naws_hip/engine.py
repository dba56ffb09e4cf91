"""Fused execution plan of the na_wsddn graph (VGG16-C5 -> RoIPoolF -> two 2-fc
branches -> WSDDN dual softmax -> entropy-gated weighted CE -> backward ->
all-reduce -> ACM momentum SGD) on one MI355X.

This is the plan the graph executor (detectron/modeling/detector.py) selects when
the recorded op list is the reference's na_wsddn graph; it runs
the same named blobs through a handful of fused HIP kernels instead of 103 ops:

  * conv body in NHWC, bias+ReLU in the implicit-GEMM epilogue (frozen, fwd only);
  * RoIPoolF fused with RoIFeatureBoost, written straight in (c,ph,pw) order;
  * fc6 of BOTH branches as ONE GEMM (N = 8192): `fc6_w` and `_[noisy]_fc6_w` are
    adjacent in one parameter arena, so the shared 400 MB `roi_feat` is read once;
  * fc7 / fc8 as batch-2 GEMMs over the interleaved [Rt, 8192] activations;
  * ReLU + Dropout in the GEMM epilogues (counter-based mask, never stored: the
    backward gate is `drop_out > 0`);
  * the whole loss tail in ~8 small kernels, per-image segments keyed on rois[:,0];
  * fc6 wgrad cut into row chunks, each all-reduced (RCCL) while the next runs;
  * one fused SGD launch over the whole parameter arena.

Blob names, shapes and update rule follow the reference:
  detectron/modeling/VGG16.py:9-48, wsl_heads.py:23-56,213-227,654-681,
  webly_heads.py:32-74,123-216,265-391,463-502, optimizer_wsl.py:52-137,
  detectron/ops/acm_weightdecay_momentum_sgd_op.h:48-112.
"""

import os

import numpy as np
import torch

from . import lib as L
from . import ops
from .reducer import ArenaReducer, message_plan, message_slice, owner_blocks, owner_pieces

VGG16_CONVS = [
    # (name, cin, cout, dilation) and pool markers
    ('conv1_1', 3, 64, 1), ('conv1_2', 64, 64, 1), ('pool', 2),
    ('conv2_1', 64, 128, 1), ('conv2_2', 128, 128, 1), ('pool', 2),
    ('conv3_1', 128, 256, 1), ('conv3_2', 256, 256, 1), ('conv3_3', 256, 256, 1), ('pool', 2),
    ('conv4_1', 256, 512, 1), ('conv4_2', 512, 512, 1), ('conv4_3', 512, 512, 1), ('pool4',),
    ('conv5_1', 512, 512, None), ('conv5_2', 512, 512, None), ('conv5_3', 512, 512, None),
]

HIDDEN = 4096

# The side streams of every engine of a process, by (device, role): 'conv0', 'conv1', ... (one
# chain per image) and 'update'.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
# round-robin as they are first used, and two streams on one queue run in order; a process that
# builds several engines one after another (bench.py's plans, the test suite) would otherwise
# keep drawing new streams from torch's pool until the image chains of one engine share a queue
# with each other or with the main stream (measured: the bf16 plan inside the default bench line
# at 242 instead of 280 img/s).  Shared streams keep the process at main + B + 2 streams; engines
# that are driven one at a time - the only way they are used - lose nothing.
_SIDE_STREAMS = {}


def side_stream(device, role):
    key = (str(device), role)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def head_param_specs(num_fg_classes, dim_in=512, roi_size=7):
    """Arena order.  Adjacent pairs form the fused GEMM operands."""
    k6, c = dim_in * roi_size * roi_size, num_fg_classes
    return [
        ('fc6_w', (HIDDEN, k6)), ('_[noisy]_fc6_w', (HIDDEN, k6)),
        ('fc6_b', (HIDDEN,)), ('_[noisy]_fc6_b', (HIDDEN,)),
        ('fc7_w', (HIDDEN, HIDDEN)), ('_[noisy]_fc7_w', (HIDDEN, HIDDEN)),
        ('fc7_b', (HIDDEN,)), ('_[noisy]_fc7_b', (HIDDEN,)),
        ('fc8c_w', (c, HIDDEN)), ('fc8d_w', (c, HIDDEN)),
        ('noisy_fc8c_w', (c, HIDDEN)), ('noisy_fc8d_w', (c, HIDDEN)),
        ('fc8c_b', (c,)), ('fc8d_b', (c,)), ('noisy_fc8c_b', (c,)), ('noisy_fc8d_b', (c,)),
    ]


class ParamArena(object):
    """One flat fp32 buffer cut into named blobs (params / grads / momentum share the cut)."""

    def __init__(self, specs, device):
        self.specs = specs
        self.offsets = {}
        off = 0
        for name, shape in specs:
            n = int(np.prod(shape))
            self.offsets[name] = (off, n, shape)
            off += n
        self.total = off
        self.device = device

    def alloc(self, zero=True):
        return (torch.zeros if zero else torch.empty)((self.total,), device=self.device,
                                                      dtype=torch.float32)

    def view(self, flat, name):
        off, n, shape = self.offsets[name]
        return flat[off:off + n].view(shape)

    def span(self, flat, first, last):
        """Contiguous slice covering blobs first..last (inclusive, arena order)."""
        o0 = self.offsets[first][0]
        o1, n1, _ = self.offsets[last]
        return flat[o0:o1 + n1]


class WsddnEngine(object):
    def __init__(self, num_classes, device, dilation=2, roi_size=7, dropout=0.5, is_mean=True,
                 momentum=0.9, weight_decay=5e-4, iter_size=1, gpu_num=1, seed=11,
                 process_group=None, world_size=1, allreduce_chunks=0, freeze_conv_body=True,
                 mfma_dtype='fp16x2', scale_momentum=True, scale_momentum_threshold=1.1,
                 sharded_update=False, rank=None, pipeline_update=None):
        if not freeze_conv_body:
            raise NotImplementedError('only TRAIN.FREEZE_CONV_BODY: True is on the hot path '
                                      '(SURVEY.md fact 2): the conv body has no backward')
        if mfma_dtype not in ('fp32', 'fp32x3', 'fp16x2', 'bf16'):
            raise ValueError("mfma_dtype must be 'fp32', 'fp32x3', 'fp16x2' or 'bf16'")
        L.load()
        # 'fp32'  : every GEMM on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak).
        # 'fp16x2' (default): fc6/fc7 forward, dgrad and wgrad on the f16 matrix cores, each fp32
        #           operand a row-scaled hi + lo pair of f16 planes, 3 MFMA passes, fp32 accumulate
        #           (csrc/gemm_x3.hip, naws_split_f16x2): as accurate as fp32x3 on every case of
        #           tests/test_gpu_h2.py / profiles/r01_x3_accuracy.md, 1.6x faster.
        # 'fp32x3': fc6/fc7 forward, dgrad and wgrad on the bf16 matrix cores with each fp32
        #           operand split exactly into three bf16 planes and six MFMA passes
        #           (csrc/gemm_x3.hip): fp32-accurate (tests/test_gpu_x3.py), ~1.65x faster.
        #           The weight planes are re-split after every SGD update, on the update stream.
        # 'bf16': conv2..conv5 and fc6/fc7 (forward, dgrad, wgrad) multiply in
        # v_mfma_f32_32x32x16_bf16 with fp32 accumulation; parameters, activations, gradients,
        # fc8, the dual softmax, the loss and the SGD update stay fp32 (BASELINE.json configs[3]:
        # "bf16 MFMA conv/fc with fp32 loss")
        self.mfma_dtype = mfma_dtype
        self.C = num_classes - 1
        self.device = device
        self.dilation = dilation
        self.roi_size = roi_size
        self.spatial_scale = 1.0 / 8.0 if dilation == 2 else 1.0 / 16.0
        self.dropout = float(dropout)
        self.is_mean = bool(is_mean)
        self.momentum = float(momentum)
        self.iter_size = int(iter_size)
        self.gpu_num = int(gpu_num)
        self.seed = int(seed)
        # SOLVER.SCALE_MOMENTUM / SCALE_MOMENTUM_THRESHOLD (detector.py:527-559)
        self.scale_momentum = bool(scale_momentum)
        self.scale_momentum_threshold = float(scale_momentum_threshold)
        self.pg, self.world_size = process_group, int(world_size)
        self.reducer = ArenaReducer(process_group, world_size)
        # NAWS.SHARDED_UPDATE (N > 1, fp16x2 plan): fc6_w's gradient rows are reduced to ONE owner
        # each (reduce-scatter), the owner updates its 8192 / N rows (fp32 master rows + momentum
        # live there only), and the updated fp32 rows + their scale words come back by all-gather;
        # every rank then splits the rows into operand planes with the owners' scales.  Same bytes
        # on the links as the all-reduce; the update's HBM traffic beside the next conv body drops
        # from 4.9 GB to 4.9 / N GB for the owned rows plus 1.6 (N-1)/N GB for the planes of the
        # rows the other ranks own (their fp32 rows read once after the gather, planes written);
        # parameters bit-identical to the all-reduce route
        # (tests/test_distributed_cpu.py, tests/test_gpu_two_ranks.py).  See _apply_update_sharded.
        self.sharded_update = bool(sharded_update)
        # NAWS.PIPELINE_UPDATE (N > 1, fp16x2 plan; None = on whenever there is an exchange): the
        # deferred update runs PIECE BY PIECE - fc6's biases, then fc6_w in two 4096-row pieces,
        # each as soon as the messages covering it have arrived, then the rest - and the next
        # iteration's fc6 forward starts each piece behind ITS update (naws_gemm_f32_f16x2_nt_cols),
        # so the tail of the exchange hides under fc6 forward too instead of stalling the head
        # (2 ranks: the one xGMI link needs ~10 ms for the 958 MB, the conv body + RoIPool hide 3).
        # Same sums, same update arithmetic, same Dropout masks: parameters bit-identical to the
        # unpipelined route (tests/test_gpu_two_ranks.py).  See _apply_update_pipelined.
        self.pipeline_update = pipeline_update
        if rank is None and process_group is not None:
            import torch.distributed as dist
            rank = dist.get_rank(process_group)
        self.rank = int(rank or 0)
        self._shard = None
        self._mom_synced = True
        # fc6_w's gradient (822 MB of the 958 MB all-reduce) can be reduced in row chunks while
        # the rest of its wgrad GEMM still runs.  0 = auto.  The collective has the tail of the
        # backward pass + the next iteration's parameter-free conv body + RoIPool (~5.7 ms) to
        # hide in; expected exchange: ~4 ms at 8 ranks (7 xGMI links per GPU, less when RCCL shares
        # the CUs with the conv kernels), ~7 ms at 4 (3 links), ~11-13 ms at 2 (one link).  Chunking
        # starts it earlier at the price of wgrad tile quantisation, measured on one GPU with
        # --force-dist: 2 chunks +0.15 ms, 4 chunks +0.4 ms per step.  Auto: 4 chunks at 2 ranks,
        # 2 chunks from 3 ranks up (cheap insurance at 8), 1 for a single rank.
        ac = int(allreduce_chunks)
        ws = int(world_size)
        self.allreduce_chunks = ac if ac >= 1 else (4 if ws == 2 else 2 if ws > 2 else 1)
        self.k6 = 512 * roi_size * roi_size
        self.ld8 = (2 * self.C + 3) // 4 * 4       # per-branch column block of the logit matrices

        self.arena = ParamArena(head_param_specs(self.C, 512, roi_size), device)
        self.params = self.arena.alloc()
        self.grads = self.arena.alloc()
        self.momentum_buf = self.arena.alloc()
        self.acmgrad = self.arena.alloc() if self.iter_size != 1 else None
        ends, lr_mult, wd = [], [], []
        for name, _ in self.arena.specs:
            off, n, _s = self.arena.offsets[name]
            is_bias = name.endswith('_b')
            # biases: no weight decay, 2x lr (optimizer_wsl.py:106-123); '_lrm10_' -> x10 (:125)
            lm = 2.0 if is_bias else 1.0
            if '_lrm10_' in name:
                lm *= 10.0
            w = 0.0 if is_bias else float(weight_decay)
            ends.append(off + n)
            if len(lr_mult) and lr_mult[-1] == lm and wd[-1] == w:
                ends.pop()
                ends[-1] = off + n          # same hyper-parameters as the previous blob: one run
                continue
            lr_mult.append(lm)
            wd.append(w)
        # the fused SGD kernel works on float4s: run boundaries must not split one (they are
        # 4096-multiples or the four fc8 biases = 4C floats, for any class count)
        if any(e % 4 for e in ends):
            raise NotImplementedError('SGD hyper-parameter runs must end on multiples of 4 floats')
        self._seg_host = (list(ends), list(lr_mult), list(wd))
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=device)
        self.seg_lr_mult = torch.tensor(lr_mult, dtype=torch.float32, device=device)
        self.seg_wd = torch.tensor(wd, dtype=torch.float32, device=device)
        self.lr = torch.zeros((1,), dtype=torch.float32, device=device)
        self._lr_host = 0.0
        self.sgd_iter_count = 0      # the SGD op's iter_count_ state
        self.step_count = 0          # forward/backward passes run (dropout stream)
        self.conv = {}               # name -> (weight (OIHW or packed), bias)
        self.stat_state = None
        # ---- the public toggles (everything else about the plan is fixed; the arms that lost
        # their A/B are gone - docs/history/ has each measurement) ---------------------------------
        #   mfma_dtype           the arithmetic plan (constructor)
        #   sharded_update       NAWS.SHARDED_UPDATE (constructor)
        #   allreduce_chunks     fc6_w wgrad / exchange row chunks (constructor)
        self.conv_streams = True     # one HIP stream per image for the conv body (bench --no-conv-streams)
        # the SGD kernel writes the updated fc6_w / fc7_w operand planes itself (fp16x2: scale from
        # twice the row maximum before the update, a device-side conditional re-split covers a row
        # that outgrows it; fp32x3 / bf16: exact / rounded planes); False: update, then re-split
        self.fused_planes = True
        # train_step() without a gradient exchange: fc6_w (86 % of the parameters) is updated in
        # the epilogue of its own wgrad GEMM - the gradient is never written (see train_step)
        self.fuse_wgrad_update = True
        # With the conv body frozen, the next iteration's conv + RoIPool do not depend on the
        # parameters: the all-reduce wait and the SGD kernel run on a side stream underneath
        # them (see sgd_step).  Same arithmetic, same order of updates; flush() joins the
        # streams (before the head, checkpoints, lr changes, end of run).
        self.defer_update = None     # None/True -> side stream; False -> inline on the main stream
        # ---- fixed choices, named where the code branches on them ------------------------------
        # Winograd for conv4_1..conv5_3 (and conv3_x in the fp32 plan); under F(2x2) the layers with
        # >= 256 halo tiles per launch (conv1_2..conv3_3 at 600x1000, the deep layers of the larger
        # TTA scales) take the direct kernel: 512 / 1024 / never measured 45.3 / 47.5 / 47.4 vs 44.6
        # ms per TTA image
        self.DIRECT_MIN_TILES = 256
        # fp16x2 plan: Winograd F(4x4,3x3) (csrc/winograd4.hip) for the Winograd layers with at
        # least this many input channels (256: conv4_1..conv5_3; 512 leaves conv4_1 to F(2x2)'s
        # fused kernel: conv body 2.099 vs 2.082 ms, tools/ab_wino4.py); 0 = F(2x2) everywhere
        self.WINO_F4_MIN_CIN = 256
        # fp32x3: up to this many output channels the direct 3-plane kernel, Winograd above
        self.X3_DIRECT_MAX_COUT = 256
        # fc8's products (tiny output, long K): K in 4 slices + a deterministic second pass (58 vs
        # 97 us, tools/bench_fc8.py)
        self.FC8_KSPLIT = 4
        # entries of VGG16_CONVS each image runs before the deferred update is queued beside the
        # chains (1: conv1_1; 2: conv1_1 + conv1_2 with pool1)
        self.UPDATE_AFTER = 2 if mfma_dtype == 'bf16' else 1
        # fp16x2 plan: each image's proposals are pooled on that image's conv stream (conv_body)
        self.ROI_POOL_ON_CHAINS = True
        self.conv_wino = {}
        self._rm_table = None
        self._fc8_ws = None
        self._w6_updated = None      # the region table for the deferred kernel once fc6_w is done
        self._sgd_regions_rest = None
        self._sgd_regions = None
        self._seg_ring = None
        self._amax5 = None
        self.conv_direct_h2 = {}
        self._streams = []
        self._update_pending = False
        self._update_waiting = False
        self._upd_stream = None
        self._upd_event = None
        self._piece_events = None    # pipelined update: [(r0, r1, event)] of the step being joined
        self._pipe = None            # its tables (built on first use)
        self._wplanes = None         # split planes of fc6_w / fc7_w / fc7_w^T (16-bit MFMA plans)
        self._planes_dirty = True
        # MEASUREMENT AID: NAWS_ENGINE_SET="DIRECT_MIN_TILES=100000,WINO_F4_MIN_CIN=0" overrides the
        # upper-case integer choices above for every engine of the process (A/B runs of entry points
        # that build their engine themselves: bench.py --infer, the CLIs)
        for kv in filter(None, os.environ.get('NAWS_ENGINE_SET', '').split(',')):
            k, v = kv.split('=')
            if not (k.isupper() and isinstance(getattr(self, k, None), int)):
                raise ValueError('NAWS_ENGINE_SET: %r is not an integer engine choice' % k)
            setattr(self, k, int(v))

    # ------------------------------------------------------------------ params
    def blob(self, name):
        self.flush()
        self._planes_dirty = True     # the caller may write through the view
        return self.arena.view(self.params, name)

    def blob_ro(self, name):
        """Read-only view of a parameter (checkpoint saves, statistics): the pending update is
        applied first, but the operand planes stay valid - a save in the middle of a run must not
        change the arithmetic of the steps after it (the planes the update kernels maintain carry
        bound-derived scales, a fresh split exact ones: both correct, not bit-identical)."""
        self.flush()
        return self.arena.view(self.params, name)

    def _one_hyper_run(self, start, count):
        """True when arena elements [start, start + count) share one (lr_mult, weight decay):
        the plane-writing SGD kernel takes ONE pair per region (head_ops.hip sgd_segment)."""
        ends = self._seg_host[0]
        i = next((k for k, e in enumerate(ends) if e > start), None)
        return i is not None and ends[i] >= start + count

    def _weight_views(self):
        w6 = self.arena.span(self.params, 'fc6_w', '_[noisy]_fc6_w').view(2 * HIDDEN, self.k6)
        w7 = self.arena.span(self.params, 'fc7_w', '_[noisy]_fc7_w').view(2, HIDDEN, HIDDEN)
        return w6, w7

    def _refresh_weight_planes(self):
        """fp16x2 / fp32x3 / bf16: re-split fc6_w / fc7_w (and fc7_w^T for the dgrad) into their
        16-bit operand planes."""
        w6, w7 = self._weight_views()
        cv = {'fp32x3': ops.split_bf16x3, 'fp16x2': ops.split_f16x2}.get(self.mfma_dtype,
                                                                         ops.to_bf16_slab)
        if self._wplanes is None and self.mfma_dtype == 'fp16x2':
            # the scale blocks of w6 / w7 live in one arena, [operand][maxima | 1/scale][8192], so
            # that the fused SGD kernel can report the updated rows' maxima straight into them
            n6 = 2 * HIDDEN
            self._wscales = torch.zeros((4 * n6,), device=self.device, dtype=torch.float32)
            f16 = dict(device=self.device, dtype=torch.float16)
            p6 = ops.F16x2(torch.empty((2, self.k6 // 16, n6, 16), **f16),
                           self._wscales[:2 * n6].view(2, n6))
            p7 = ops.F16x2(torch.empty((2, 2, HIDDEN // 16, HIDDEN, 16), **f16),
                           self._wscales[2 * n6:].view(2, 2, HIDDEN))
            self._wplanes = dict(w6=cv(w6, out=p6), w7=cv(w7, out=p7), w7t=cv(w7, transpose=True))
            o6, o7 = self.arena.offsets['fc6_w'][0], self.arena.offsets['fc7_w'][0]
            self._rm_table = ops.RowmaxTable([(o6, o6 + n6 * self.k6, self.k6, 0),
                                              (o7, o7 + n6 * HIDDEN, HIDDEN, 2 * n6)], self.device)
            one_run = (self._one_hyper_run(o6, n6 * self.k6) and self._one_hyper_run(o7, n6 * HIDDEN))
            if self.k6 % 256 == 0 and one_run:
                # (a configuration that gives the two branches different lr_mult / weight decay
                # keeps the element-wise kernel + re-split: _sgd_regions stays None)
                # [w6 rows | w7 rows]: the maxima before an update (the scale bounds of the planes
                # the update writes) and the "a row outgrew its bound" word
                self._wbound = torch.zeros((2 * n6,), device=self.device, dtype=torch.int32)
                self._wovf = torch.zeros((1,), device=self.device, dtype=torch.int32)
                sc = self._wscales.view(2, 2, n6)            # [operand][maxima | 1/scale][rows]
                # fc7_w's COLUMN maxima = the row maxima of fc7_w^T (the dgrad's operand): the
                # kernel reports them too, so the transposed planes need no maxima pass either
                cm7 = self._wplanes['w7t'].scales[0].view(torch.int32)
                self._sgd_regions = ops.SgdPlaneRegions([
                    (o6, n6, self.k6, n6, p6.planes, self._wbound[:n6], sc[0, 0].view(torch.int32),
                     sc[0, 1]),
                    (o7, n6, HIDDEN, HIDDEN, p7.planes, self._wbound[n6:], sc[1, 0].view(torch.int32),
                     sc[1, 1], cm7)])
                # the same table with fc6_w marked "updated elsewhere" (train_step)
                self._sgd_regions_rest = ops.SgdPlaneRegions([
                    (o6, n6, self.k6, n6, None, None, None, None),
                    (o7, n6, HIDDEN, HIDDEN, p7.planes, self._wbound[n6:], sc[1, 0].view(torch.int32),
                     sc[1, 1], cm7)])
        elif self._wplanes is None:
            self._wplanes = dict(w6=cv(w6), w7=cv(w7), w7t=cv(w7, transpose=True))
            n6 = 2 * HIDDEN
            o6, o7 = self.arena.offsets['fc6_w'][0], self.arena.offsets['fc7_w'][0]
            if (self.k6 % 256 == 0 and self._one_hyper_run(o6, n6 * self.k6)
                    and self._one_hyper_run(o7, n6 * HIDDEN)):
                # fp32x3 / bf16: the SGD kernel writes these planes too (exact / rounded bf16:
                # no scales, nothing to bound)
                fmt = L.PLANES_BF16X3 if self.mfma_dtype == 'fp32x3' else L.PLANES_BF16
                self._sgd_regions = ops.SgdPlaneRegions([
                    (o6, n6, self.k6, n6, self._wplanes['w6'], None, None, None),
                    (o7, n6, HIDDEN, HIDDEN, self._wplanes['w7'], None, None, None)], fmt)
                # the same table with fc6_w marked "updated elsewhere" (train_step, bf16 plan)
                self._sgd_regions_rest = ops.SgdPlaneRegions([
                    (o6, n6, self.k6, n6, None, None, None, None),
                    (o7, n6, HIDDEN, HIDDEN, self._wplanes['w7'], None, None, None)], fmt)
        else:
            cv(w6, out=self._wplanes['w6'])
            cv(w7, out=self._wplanes['w7'])
            cv(w7, transpose=True, out=self._wplanes['w7t'])
        self._planes_dirty = False

    def grad_blob(self, name):
        return self.arena.view(self.grads, name)

    def momentum_blob(self, name):
        self.flush()
        return self.arena.view(self.momentum_buf, name)

    def set_conv_blobs(self, blobs):
        """blobs: name_w [O,I,3,3], name_b [O] (reference layout).  Packs once."""
        for item in VGG16_CONVS:
            if item[0].startswith('pool'):
                continue
            name, dil = item[0], item[3]
            w = blobs[name + '_w'].to(self.device, torch.float32).contiguous()
            b = blobs[name + '_b'].to(self.device, torch.float32).contiguous()
            use_wino = (self.mfma_dtype != 'bf16' and w.shape[1] >= 128 and w.shape[0] >= 256)
            if self.mfma_dtype == 'fp32x3' and w.shape[0] <= self.X3_DIRECT_MAX_COUT:
                use_wino = False          # conv3_x: the 3-plane direct kernel fills the chip there
            x3conv = (self.mfma_dtype in ('fp32x3', 'fp16x2') and name != 'conv1_1' and not use_wino)
            if name == 'conv1_1':
                packed = w
                # |conv1_1(x)| <= max|x| * max_c sum|w_c| + max|b|: the operand-scale bound of
                # the layer that consumes it, without a pass over its 154 MB output
                self._c11_bound = (float(w.abs().sum(dim=(1, 2, 3)).max().item()),
                                   float(b.abs().max().item()))
            elif (x3conv and self.mfma_dtype == 'fp16x2' and dil == 1
                  and w.shape[0] <= 128 and w.shape[0] % 32 == 0 and w.shape[1] % 16 == 0):
                # f16 hi / lo planes [2][9*Cin/16][Cout][16] + per-channel scales
                packed = ops.split_f16x2(ops.conv3x3_pack_weight(w).view(w.shape[0], -1))
            elif x3conv:
                # weight planes [3][9*Cin/16][Cout][16] of the packed [Cout][3][3][Cin] weight
                packed = ops.split_bf16x3(ops.conv3x3_pack_weight(w).view(w.shape[0], -1))
            elif use_wino:
                # (conv3_2 / conv3_3, 256 -> 256 at 150 x 250, stay on the direct kernel: as F(4x4)
                # their V and M are 88 MB each way for a quarter of conv4_2's products - 0.24 vs
                # 0.19 ms per layer-image inside a step)
                f4 = (self.mfma_dtype == 'fp16x2' and 0 < self.WINO_F4_MIN_CIN <= w.shape[1]
                      and w.shape[1] % 32 == 0 and w.shape[0] >= 512)
                packed = (ops.winograd4_weight_transform(w) if f4          # [36][Cout][Cin]
                          else ops.winograd_weight_transform(w))           # [16][Cout][Cin]
                if self.mfma_dtype == 'fp16x2':
                    packed = ops.split_f16x2(packed)           # F16x2, planes [2][16][Cin/16][Cout][16]
                    # (F(4x4) layers never switch: 42.4 vs 43.5 ms per TTA image with the direct
                    # form taking over at the large scales, 44.5 for F(2x2) with it)
                    if not f4 and w.shape[0] % 128 == 0 and \
                            (dil == 1 or (dil is None and self.dilation in (1, 2))):
                        # also the direct form: chosen per input size in _conv_chain
                        self.conv_direct_h2[name] = ops.split_f16x2(
                            ops.conv3x3_pack_weight(w).view(w.shape[0], -1))
                elif self.mfma_dtype == 'fp32x3':
                    packed = ops.split_bf16x3(packed)          # planes [3][16][Cin/16][Cout][16]
            elif self.mfma_dtype == 'bf16' and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                # one bf16 plane [9*Cin/16][Cout][16]: the wave-private halo-tile kernel
                packed = ops.to_bf16_slab(ops.conv3x3_pack_weight(w).view(w.shape[0], -1))
            else:
                packed = ops.conv3x3_pack_weight(w)            # [Cout][3][3][Cin]
            self.conv[name] = (packed, b, w)
            self.conv_wino[name] = use_wino

    def set_head_blobs(self, blobs):
        for name, shape in self.arena.specs:
            if name in blobs:
                self.blob(name).copy_(blobs[name].to(self.device, torch.float32).view(shape))

    def broadcast_parameters(self, src=0):
        """COLLECTIVE (every rank): rank `src`'s parameters, momentum and conv body to every rank
        (reference: utils/net_wsl.py:183-207 copies GPU 0's blobs to the other GPUs through the
        host; here one RCCL broadcast per buffer).  The conv weights are re-packed from what
        arrived, the operand planes re-split on the next forward.  No-op without a process group."""
        if self.pg is None:
            return
        import torch.distributed as dist
        self.flush()
        root = dist.get_global_rank(self.pg, src)
        dist.broadcast(self.params, root, group=self.pg)
        dist.broadcast(self.momentum_buf, root, group=self.pg)
        if self.conv:
            for _wp, b, w in self.conv.values():
                dist.broadcast(w, root, group=self.pg)
                dist.broadcast(b, root, group=self.pg)
            self.set_conv_blobs({k: v for k, v in self.export_blobs(False).items()
                                 if k.startswith('conv')})
        # self.params was written directly: the operand planes no longer match it
        self._planes_dirty = True

    def state_tensors(self):
        """Every buffer that must be bit-identical on all ranks of a data-parallel job after an
        update: parameters, momentum, and (16-bit MFMA plans) the operand planes of fc6_w / fc7_w /
        fc7_w^T with their scale words.  The pending update is joined first; under
        NAWS.SHARDED_UPDATE call gather_sharded_state() (collective) before comparing momentum."""
        self.flush()
        out = dict(params=self.params, momentum=self.momentum_buf)
        if self._wplanes is not None and not self._planes_dirty:
            for k, v in self._wplanes.items():
                if isinstance(v, ops.F16x2):
                    out['planes_' + k], out['scales_' + k] = v.planes, v.scales
                else:
                    out['planes_' + k] = v
        return out

    def set_update_route(self, pipeline_update=None, sharded_update=False):
        """COLLECTIVE when leaving the sharded route (the owners' momentum rows are gathered
        first).  Switch how the deferred update of an N > 1 step runs, at a step boundary:
        pipeline_update None = the default (piece by piece whenever there is an exchange), False =
        one launch behind the whole exchange; sharded_update = NAWS.SHARDED_UPDATE.  All three
        routes hold bit-identical state across the ranks, so a job may change route mid-run
        (bench.py's in-run A/B, the fallback of a training job)."""
        self.flush()
        if self._shard_blocks() is not None:
            self.gather_sharded_state()
        self.pipeline_update = pipeline_update
        self.sharded_update = bool(sharded_update)
        self._shard, self._mom_synced = None, True
        self._pipe_sent = False

    def export_blobs(self, with_momentum=True):
        self.flush()
        if with_momentum and not self._mom_synced:
            raise RuntimeError('NAWS.SHARDED_UPDATE: fc6_w momentum rows live with their owners; '
                               'call gather_sharded_state() on EVERY rank before a checkpoint')
        out = {}
        for name, (wp, b, w) in self.conv.items():
            out[name + '_w'], out[name + '_b'] = w, b
        for name, _ in self.arena.specs:
            out[name] = self.blob_ro(name)
            if with_momentum:
                out[name + '_momentum'] = self.momentum_blob(name)
        return out

    # ---------------------------------------------------------------- forward
    def _conv_chain(self, data, out=None, amax_final=None, first=0, end=None, x=None,
                    bound_in=None, amax_last=None, affine_in=None):
        """data NCHW [b,3,H,W] -> conv5_3 NHWC, on the current stream.  amax_final (int32 [1],
        fp16x2 plan): receives the bit pattern of max|conv5_3| for the RoIPool operand scale.
        first / end: run only VGG16_CONVS[first:end] (x = the input of layer `first`, bound_in = the
        word holding an upper bound of max|x|); amax_last: a pre-zeroed word that the LAST layer of
        the range max-es its max|y| into (shared by the per-image chains)."""
        last = VGG16_CONVS[-1][0]
        end = len(VGG16_CONVS) if end is None else end
        # fp16x2 Winograd layers hand max|y| to the next layer (its operand scale needs an upper
        # bound of max|x|; a max-pool in between only lowers it), saving that layer's own pass
        amax = (torch.zeros((len(VGG16_CONVS) + 1,), device=self.device, dtype=torch.int32)   # one fill
                if self.mfma_dtype == 'fp16x2' else None)
        prev, affine = None, (1.0, 0.0)  # slot holding the bound for the current x, if any
        if bound_in is not None:         # (affine_in: the bound is bound_in * mul + add, once)
            amax[-1:].copy_(bound_in)
            prev = len(VGG16_CONVS)
            if affine_in is not None:
                affine = affine_in
        affine_slot = 0 if affine_in is None else len(VGG16_CONVS)
        fused_pool = False               # the previous layer's epilogue already pooled
        last_conv = max(i for i in range(first, end) if not VGG16_CONVS[i][0].startswith('pool'))
        for li, item in enumerate(VGG16_CONVS):
            if li < first or li >= end:
                continue
            if item[0] == 'pool':
                if not fused_pool and not (li == first and self._pool_done):
                    x = ops.maxpool2x2_nhwc(x, 2)
                fused_pool = False
            elif item[0] == 'pool4':
                x = ops.maxpool2x2_nhwc(x, 1 if self.dilation == 2 else 2)
            else:
                name, _, _, dil = item
                wp, b, _w = self.conv[name]
                if name == 'conv1_1':
                    x = ops.conv3x3_c3_nchw_to_nhwc(data, wp, b, True)
                    prev = None
                    if self.mfma_dtype == 'fp16x2':
                        ops.amax_word(data, out=amax[li:li + 1])     # 7 MB: the network input
                        prev, affine = li, self._c11_bound
                else:
                    d = dil if dil is not None else (2 if self.dilation == 2 else 1)
                    dst = out if name == last else None
                    wd = self.conv_direct_h2.get(name)
                    if wd is not None and d in (1, 2):
                        # direct 2 x f16 halo-tile kernel where it fills the chip (>= one 8x32-pixel
                        # x 128-channel tile per CU: conv3_x at 600x1000, and every deep layer once
                        # the images share a launch), Winograd below that
                        tiles = x.shape[0] * ((x.shape[1] + 7) // 8) * ((x.shape[2] + 31) // 32)
                        if tiles * (wd.planes.shape[-2] // 128) >= self.DIRECT_MIN_TILES:
                            wp = wd
                    if isinstance(wp, ops.F16x2):
                        bound = None if prev is None else amax[prev:prev + 1]
                        word = amax_final if (name == last and amax_final is not None) \
                            else amax[li:li + 1]
                        zeroed = word is not amax_final
                        if li == last_conv and amax_last is not None:
                            word, zeroed = amax_last, True
                        if wp.planes.dim() == 4:         # direct halo-tile kernel
                            mul, add = affine if prev == affine_slot else (1.0, 0.0)
                            # a 2x2 / stride-2 max-pool that follows is taken in the epilogue
                            fused_pool = (li + 1 < len(VGG16_CONVS)
                                          and VGG16_CONVS[li + 1][0] == 'pool')
                            x = ops.conv3x3_nhwc_f16x2(x, wp, b, True, out=dst, amax_in=bound,
                                                       in_mul=mul, in_add=add, amax_out=word,
                                                       pool2=fused_pool, amax_out_zeroed=zeroed,
                                                       dilation=d)
                        else:                            # Winograd, f16 batch GEMMs
                            x = ops.conv3x3_winograd_nhwc_f16x2(x, wp, b, d, True, out=dst,
                                                                amax_in=bound, amax_out=word)
                        prev = li
                    elif self.mfma_dtype == 'bf16' and wp.dtype == torch.bfloat16:
                        fused_pool = (d == 1 and dst is None and li + 1 < len(VGG16_CONVS)
                                      and VGG16_CONVS[li + 1][0] == 'pool')
                        x = ops.conv3x3_nhwc_bf16_wp(x, wp, b, d, True, out=dst, pool2=fused_pool)
                        prev = None
                    else:
                        if wp.dtype == torch.bfloat16:
                            conv = (ops.conv3x3_winograd_nhwc_f32x3 if self.conv_wino[name]
                                    else ops.conv3x3_nhwc_f32x3)
                        elif self.mfma_dtype == 'bf16':
                            conv = ops.conv3x3_nhwc_bf16
                        else:
                            conv = (ops.conv3x3_winograd_nhwc if self.conv_wino[name]
                                    else ops.conv3x3_nhwc)
                        if (conv is ops.conv3x3_nhwc_f32x3 and d == 1 and dst is None
                                and li + 1 < len(VGG16_CONVS) and VGG16_CONVS[li + 1][0] == 'pool'
                                and wp.shape[-2] % 64 == 0 and wp.shape[-2] <= 256):
                            x = conv(x, wp, b, d, True, pool2=True)     # pool1..3 in the epilogue
                            fused_pool = True
                        else:
                            x = conv(x, wp, b, d, True, out=dst)
                        prev = None
        self._pool_done = fused_pool     # a range that ends on a fused pool: the next one skips it
        return x

    def conv_body(self, data, roi_job=None):
        """data NCHW [B,3,H,W] -> conv5_3 NHWC [B,H/8-1,W/8-1,512].

        roi_job = (rois, obn_scores, seg) (fp16x2 plan, one chain per image): every image's
        proposals are pooled at the tail of ITS chain, on its stream, beside the other images'
        last layers - RoIPoolF + boost straight into fc6's operand planes, rows seg[i]..seg[i+1];
        the finished operand is then in self._roi_operand (consumed by _roi_features).

        The 13 layers of one image are a dependent chain, and the deep layers have only a
        couple of workgroup-tiles per CU, so every layer ends on a partially filled chip.
        Images are independent: each image's chain is queued on its own HIP stream and the
        hardware packs the tail of one image's layer with the head of the other's."""
        n = data.shape[0]
        # fp16x2: max|conv5_3| per image (per chain) for the RoIPool -> fc6 operand scale
        planes = (self.mfma_dtype == 'fp16x2' and self.k6 % 32 == 0
                  and isinstance(self.conv[VGG16_CONVS[-1][0]][0], ops.F16x2))
        per_image = n > 1 and self.conv_streams
        self._pool_done = False
        self._roi_maps = None
        self._roi_operand = None
        self._amax5 = (torch.empty((n if per_image else 1,), device=self.device,
                                   dtype=torch.int32) if planes else None)
        heads_first = self.mfma_dtype in ('fp16x2', 'bf16')
        if self._update_waiting and (not per_image or not heads_first):
            self._launch_update(())        # no per-image conv1_1 head to put it behind
        if not per_image:
            res = self._conv_chain(data, amax_final=self._amax5)
            self._roi_maps_of = (res.data_ptr(), res._version)
            return res
        h, w = data.shape[2], data.shape[3]
        for _ in range(3):
            h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
        if self.dilation == 2:
            h, w = h - 1, w - 1
        else:
            h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
        out = torch.empty((n, h, w, 512), device=self.device, dtype=torch.float32)
        # (bf16 plan: RoIPoolF writes fc6's one-plane operand over the same maps)
        slab = self.mfma_dtype == 'bf16' and self.k6 % 64 == 0
        self._roi_maps = (torch.empty_like(out), torch.empty_like(out)) if (planes or slab) else None
        # the maps belong to THIS tensor in THIS state (_take_roi_maps)
        self._roi_maps_of = (out.data_ptr(), out._version)
        pooled = None
        if (roi_job is not None and planes and self.ROI_POOL_ON_CHAINS
                and roi_job[0].shape[0] > 0 and len(roi_job[2]) == n + 1):
            self._check_rows_grouped_by_image(roi_job[0], roi_job[2])
            pooled = ops.roi_pool_operand(roi_job[0].shape[0], self.k6, self.device)
        main = torch.cuda.current_stream(self.device)
        start = main.record_event()
        while len(self._streams) < n:
            self._streams.append(side_stream(self.device, 'conv%d' % len(self._streams)))
        split = (self._update_waiting and heads_first)
        h2 = self.mfma_dtype == 'fp16x2'     # operand-scale bounds travel along the chain
        if split:
            # the head of every image's chain first (conv1_1: HBM-bound, 0.06 ms alone), THEN the
            # deferred parameter update, then the rest of the chains: started together, the SGD
            # kernel's workgroups fill the CUs and the two conv1_1 launches at the head of the
            # dependent chains took 0.7 ms each (kernel trace)
            heads, evs = [], []
            for i in range(n):
                st = self._streams[i]
                st.wait_event(start)
                with torch.cuda.stream(st):
                    if self.UPDATE_AFTER <= 1:
                        wp, b, _w = self.conv['conv1_1']
                        y = ops.conv3x3_c3_nchw_to_nhwc(data[i:i + 1], wp, b, True)
                        bound, aff = ((ops.amax_word(data[i:i + 1]), self._c11_bound) if h2
                                      else (None, None))
                    else:
                        # ... and conv1_2 (+ pool1): the other HBM-heavy layer of the chain
                        bound = (torch.zeros((1,), device=self.device, dtype=torch.int32) if h2
                                 else None)
                        y = self._conv_chain(data[i:i + 1], first=0, end=self.UPDATE_AFTER,
                                             amax_last=bound)
                        aff = None
                    heads.append((y, bound, aff))
                    evs.append(st.record_event())
            pool_done = self._pool_done
            self._launch_update(evs)
        for i in range(n):
            st = self._streams[i]
            if not split:
                st.wait_event(start)
            with torch.cuda.stream(st):
                af = None if self._amax5 is None else self._amax5[i:i + 1]
                if split:
                    self._pool_done = pool_done
                    self._conv_chain(None, out=out[i:i + 1], amax_final=af,
                                     first=max(1, self.UPDATE_AFTER), x=heads[i][0],
                                     bound_in=heads[i][1], affine_in=heads[i][2])
                else:
                    self._conv_chain(data[i:i + 1], out=out[i:i + 1], amax_final=af)
                if self._roi_maps is not None:
                    # RoIPoolF's block-maxima maps of this image, beside the other image's tail
                    ops.roi_maxmaps(out[i:i + 1], self._roi_maps[0][i:i + 1], self._roi_maps[1][i:i + 1])
                    if pooled is not None and roi_job[2][i + 1] > roi_job[2][i]:
                        # ... and its proposals pooled right behind them (batch index i: the maps
                        # and conv5_3 of the other images are not touched)
                        ops.roi_pool_f_f16x2_range(out, roi_job[0], self._amax5, pooled, roi_job[2][i],
                                                   roi_job[2][i + 1], self._roi_maps, self.roi_size,
                                                   self.roi_size, self.spatial_scale,
                                                   boost=roi_job[1].reshape(-1))
                done = st.record_event()
            main.wait_event(done)
        if pooled is not None:
            # the operand belongs to exactly these tensors in exactly this state (_roi_features)
            self._roi_operand = (pooled, out.data_ptr(), self._roi_job_key(roi_job[0], roi_job[1]))
        return out

    @staticmethod
    def _roi_job_key(rois, obn_scores):
        return (rois.data_ptr(), rois._version, obn_scores.data_ptr(), obn_scores._version)

    def _check_rows_grouped_by_image(self, rois, seg):
        """Precondition of the per-image pooling on the conv streams: rows seg[i]..seg[i+1] of
        `rois` carry batch index i (the kernel takes the image from rois[:,0], the stream from
        seg).  Checked on the device, with a host sync, under NAWS_DEBUG_CHECKS=1 only."""
        import os
        if os.environ.get('NAWS_DEBUG_CHECKS') != '1':
            return
        want = torch.repeat_interleave(
            torch.arange(len(seg) - 1, device=rois.device, dtype=rois.dtype),
            torch.tensor([seg[i + 1] - seg[i] for i in range(len(seg) - 1)], device=rois.device))
        if want.numel() != rois.shape[0] or not bool((rois[:, 0] == want).all().item()):
            raise ValueError('rois must be grouped by image in order: rows seg[i]..seg[i+1] must '
                             'carry batch index i (seg = %r)' % (list(seg),))

    def _seg_to_device(self, seg):
        """Per-image row offsets -> int32 device tensor WITHOUT stalling the host: a copy from
        pageable memory (torch.tensor(list, device=...)) is synchronous and waits for everything
        queued on the stream before it, i.e. once per step the host stopped running ahead of the GPU
        and queued the next conv body just in time (the second image's chain started 1.5 ms after
        the first one's in the kernel trace).  Here: a small ring of pinned slots, async copies."""
        n = len(seg)
        if self._seg_ring is None or self._seg_ring[0][0].numel() < n:
            self._seg_ring = [[torch.empty((max(n, 64),), dtype=torch.int32).pin_memory(), None]
                              for _ in range(8)]
            self._seg_next = 0
        slot = self._seg_ring[self._seg_next]
        self._seg_next = (self._seg_next + 1) % len(self._seg_ring)
        if slot[1] is not None:
            slot[1].synchronize()          # eight copies ago: long done
        slot[0][:n] = torch.tensor(seg, dtype=torch.int32)
        out = slot[0][:n].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.current_stream(self.device).record_event()
        return out

    @staticmethod
    def segments(rois, n_img):
        """seg_off (host list) from rois[:,0]; rows must be grouped by image in order."""
        b = rois[:, 0].to(torch.int64)
        counts = torch.bincount(b, minlength=n_img).cpu().tolist()
        seg = [0]
        for c in counts:
            seg.append(seg[-1] + c)
        return seg

    def _seed(self, layer):
        return (self.seed * 0x9E3779B1 + self.step_count * 1000003 + layer * 7919) & ((1 << 62) - 1)

    def _take_roi_maps(self, conv5):
        """The block-maxima maps conv_body() built beside the chains, if `conv5` is the very
        tensor it returned, unmodified (same storage, same version counter); None for any other
        tensor of the same shape - the pooling wrapper then builds maps of what it is given."""
        maps, self._roi_maps = getattr(self, '_roi_maps', None), None
        if maps is not None and getattr(self, '_roi_maps_of', None) != (conv5.data_ptr(), conv5._version):
            maps = None
        return maps

    def _roi_features(self, conv5, rois, obn_scores):
        """RoIPoolF + boost -> the fc6 input: fp32 [Rt, k6], or (fp16x2 plan) the GEMM operand
        planes written by the pooling kernel itself."""
        done = getattr(self, '_roi_operand', None)
        self._roi_operand = None
        if done is not None and done[1] == conv5.data_ptr() \
                and done[2] == self._roi_job_key(rois, obn_scores) \
                and getattr(self, '_roi_maps_of', None) == (conv5.data_ptr(), conv5._version):
            self._roi_maps = None
            return done[0]                    # pooled per image at the tails of the conv chains
        if self._amax5 is not None:
            mine = getattr(self, '_roi_maps_of', None) == (conv5.data_ptr(), conv5._version)
            maps = self._take_roi_maps(conv5)
            if not mine:
                # a caller's own feature map (or conv_body's, modified): max|conv5_3| - the
                # bound behind the operand scale of the planes - is taken from what was passed
                self._amax5 = torch.cat([ops.amax_word(conv5[i:i + 1].contiguous())
                                         for i in range(self._amax5.numel())]) \
                    if self._amax5.numel() == conv5.shape[0] else ops.amax_word(conv5.contiguous())
            return ops.roi_pool_f_f16x2(conv5, rois, self._amax5, self.roi_size, self.roi_size,
                                        self.spatial_scale, boost=obn_scores.reshape(-1),
                                        hier=True, maps=maps)
        if self.mfma_dtype == 'bf16' and self.k6 % 64 == 0 and conv5.shape[-1] % 64 == 0:
            maps = self._take_roi_maps(conv5)
            if maps is None:
                maps = (torch.empty_like(conv5), torch.empty_like(conv5))
                ops.roi_maxmaps(conv5, maps[0], maps[1])
            return ops.roi_pool_f_bf16_slab(conv5, rois, maps, self.roi_size, self.roi_size,
                                            self.spatial_scale, boost=obn_scores.reshape(-1))
        roi_feat = ops.roi_pool_f(conv5, rois, self.roi_size, self.roi_size, self.spatial_scale,
                                  boost=obn_scores.reshape(-1), layout='NHWC', hier=True)
        return roi_feat.view(rois.shape[0], self.k6)

    def head_forward(self, roi_feat, train, both_branches=True, pieces=None):
        """roi_feat [Rt, k6] (or its F16x2 operand form) -> H6, H7 [Rt, nb*4096], logits L
        [Rt, nb*2C].  pieces: see _join_update_for_head (the pipelined update's events)."""
        f16p = isinstance(roi_feat, ops.F16x2)
        # (the bf16 plan's operand form, written by the pooling kernel: bf16 [k6/16, Rt, 16])
        slab_x = not f16p and roi_feat.dtype == torch.bfloat16 and roi_feat.dim() == 3
        rt = roi_feat.planes.shape[-2] if f16p else (roi_feat.shape[1] if slab_x else roi_feat.shape[0])
        nb = 2 if both_branches else 1
        C = self.C
        w6 = self.arena.span(self.params, 'fc6_w', '_[noisy]_fc6_w').view(2 * HIDDEN, self.k6)
        b6 = self.arena.span(self.params, 'fc6_b', '_[noisy]_fc6_b')
        w7 = self.arena.span(self.params, 'fc7_w', '_[noisy]_fc7_w').view(2, HIDDEN, HIDDEN)
        b7 = self.arena.span(self.params, 'fc7_b', '_[noisy]_fc7_b').view(2, HIDDEN)
        w8 = self.arena.span(self.params, 'fc8c_w', 'noisy_fc8d_w').view(2, 2 * C, HIDDEN)
        b8 = self.arena.span(self.params, 'fc8c_b', 'noisy_fc8d_b').view(2, 2 * C)
        drop = train and self.dropout > 0
        epi = L.EPI_BIAS_RELU_DROP if drop else L.EPI_BIAS_RELU
        bf = self.mfma_dtype == 'bf16'
        x3 = self.mfma_dtype == 'fp32x3'
        h2 = self.mfma_dtype == 'fp16x2'
        if pieces is not None and not (h2 and nb == 2):
            self._finish_pieces(pieces)      # (a caller's own plan: join the whole update first)
            pieces = None
        if self.mfma_dtype != 'fp32' and self._planes_dirty:
            self._refresh_weight_planes()
        # the activation operand in the plan's form (its own kernels, outside the timed launch)
        if h2:
            xp = roi_feat if isinstance(roi_feat, ops.F16x2) else ops.split_f16x2(roi_feat)
        elif x3:
            xp = roi_feat if roi_feat.dtype == torch.bfloat16 else ops.split_bf16x3(roi_feat)
        elif bf:
            xp = roi_feat if slab_x else ops.to_bf16_slab(roi_feat)
        tev = getattr(self, 'timing_events', None)
        if tev is not None and pieces is None:   # bench.py: HIP events around the dominant kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if h2 and pieces is not None:
            # fc6 forward piece by piece: piece (r0, r1) = weight rows / h6 columns r0..r1 of one
            # branch, launched behind the event of ITS update; same kernel, same K order, same
            # Dropout counters as the one launch below (naws_gemm_f32_f16x2_nt_cols)
            self._new_amax_arena(rt, nb)
            sc6n = self._scales(nb, rt)
            sc6t = self._scales(nb, HIDDEN) if train else None
            h6 = torch.empty((rt, nb * HIDDEN), device=self.device, dtype=torch.float32)
            main = torch.cuda.current_stream(self.device)
            rowwords = ops.amax_words(sc6n)                       # [nb, Rt]
            colwords = None if sc6t is None else ops.amax_words(sc6t).view(-1)
            for r0, r1, ev in pieces['fc6']:
                main.wait_event(ev)
                if r0 // HIDDEN != (r1 - 1) // HIDDEN:
                    raise RuntimeError('a forward piece must lie inside one branch')
                if tev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ops.gemm_f32_f16x2_nt_cols(
                    xp, self._wplanes['w6'].rows(r0, r1), h6[:, r0:r1], r0, nb * HIDDEN,
                    epilogue=epi, bias=b6[r0:r1], drop_ratio=self.dropout if drop else 0.0,
                    seed=self._seed(6), rowmax=rowwords[r0 // HIDDEN], rowmax_seg=HIDDEN,
                    colmax=None if colwords is None else colwords[r0:r1])
                if tev is not None:
                    e1.record()
                    tev.append((e0, e1))
            self._finish_pieces(pieces)          # fc7 / fc8 read the rest of the update
            tev = None
        elif h2:
            # the GEMM epilogue reports max|h6| per row of each branch (fc7's operand scale) and,
            # in training, per column (fc7 wgrad's): no pass over h6 for the maxima
            self._new_amax_arena(rt, nb)
            sc6n = self._scales(nb, rt)
            sc6t = self._scales(nb, HIDDEN) if train else None
            h6 = ops.gemm_f32_f16x2_nt(xp, self._wplanes['w6'].rows(0, nb * HIDDEN), epilogue=epi,
                                       bias=b6, drop_ratio=self.dropout if drop else 0.0,
                                       seed=self._seed(6), rowmax=ops.amax_words(sc6n),
                                       rowmax_seg=HIDDEN,
                                       colmax=None if sc6t is None else ops.amax_words(sc6t))
        elif x3:
            h6 = ops.gemm_f32x3_nt(xp, self._wplanes['w6'][:, :, :nb * HIDDEN], epilogue=epi,
                                   bias=b6, drop_ratio=self.dropout if drop else 0.0,
                                   seed=self._seed(6))
        elif bf:
            h6 = ops.gemm_bf16_slab_nt(xp, self._wplanes['w6'][:, :nb * HIDDEN], epilogue=epi, bias=b6,
                                       drop_ratio=self.dropout if drop else 0.0, seed=self._seed(6))
        else:
            h6 = ops.gemm(roi_feat, w6[:nb * HIDDEN], False, True, epilogue=epi, bias=b6,
                          drop_ratio=self.dropout if drop else 0.0, seed=self._seed(6))
        if tev is not None:
            e1.record()
            tev.append((e0, e1))
        xp = None
        h6v = h6.view(rt, nb, HIDDEN).permute(1, 0, 2)       # [nb, Rt, 4096] strided views
        h7 = torch.empty((rt, nb * HIDDEN), device=self.device, dtype=torch.float32)
        h7v = h7.view(rt, nb, HIDDEN).permute(1, 0, 2)
        if h2:
            # one pass over h6 writes both operand forms (rows for fc7, columns for fc7's wgrad)
            h6n, self._h6t = ops.split_f16x2_dual(h6v, sc6n, sc6t)
            ops.gemm_f32_f16x2_nt(h6n, self._wplanes['w7'].batches(nb), out=h7v,
                                  epilogue=epi, bias=b7, drop_ratio=self.dropout if drop else 0.0,
                                  seed=self._seed(7))
            del h6n
        elif x3:
            ops.gemm_f32x3_nt(ops.split_bf16x3(h6v), self._wplanes['w7'][:, :nb], out=h7v,
                              epilogue=epi, bias=b7, drop_ratio=self.dropout if drop else 0.0,
                              seed=self._seed(7))
        elif bf:
            ops.gemm_bf16_slab_nt(ops.to_bf16_slab(h6v), self._wplanes['w7'][:nb], out=h7v,
                                  epilogue=epi, bias=b7, drop_ratio=self.dropout if drop else 0.0,
                                  seed=self._seed(7))
        else:
            ops.gemm(h6v, w7[:nb], False, True, out=h7v, epilogue=epi, bias=b7,
                     drop_ratio=self.dropout if drop else 0.0, seed=self._seed(7))
        # logits [Rt, nb * ld8]: branch b holds fc8c | fc8d in columns b*ld8 .. b*ld8 + 2C.  ld8 = 2C
        # rounded up to 4 floats so the batch-2 GEMM operands stay 16-byte aligned for any class
        # count (odd C: the weights / bias are copied into zero-padded [ld8, 4096] operands)
        ld8 = self.ld8
        w8g, b8g = self._fc8_operands(w8, b8)
        lg = torch.empty((rt, nb * ld8), device=self.device, dtype=torch.float32)
        lgv = lg.view(rt, nb, ld8).permute(1, 0, 2)
        self._fc8_gemm(h7v, w8g[:nb], False, True, lgv, L.EPI_BIAS, b8g)
        return h6, h7, lg

    def _fc8_gemm(self, a, b, trans_a, trans_b, out, epilogue=L.EPI_NONE, bias=None):
        """fc8's products have a small output and a long inner dimension (K = 4096 forward,
        K = proposals for dW8): K in FC8_KSPLIT slices + a deterministic second pass
        (ops.gemm_splitk: 58 vs 97 us on the bench shape, tools/bench_fc8.py); 0 = one pass."""
        ks = int(self.FC8_KSPLIT)
        if ks <= 1:
            return ops.gemm(a, b, trans_a, trans_b, out=out, epilogue=epilogue, bias=bias)
        need = out.shape[-2] * out.shape[-1] * (out.shape[0] if out.dim() == 3 else 1) * ks
        if self._fc8_ws is None or self._fc8_ws.numel() < need:
            self._fc8_ws = torch.empty((need,), device=self.device, dtype=torch.float32)
        return ops.gemm_splitk(a, b, trans_a, trans_b, out=out, epilogue=epilogue, bias=bias,
                               ksplit=ks, workspace=self._fc8_ws)

    def _new_amax_arena(self, rt, nb):
        """One zero-fill per step for every |.| maxima vector the GEMM epilogues accumulate into
        (h6 rows + columns, dZ7 rows + columns, dZ6 columns)."""
        need = 2 * (2 * nb * rt + 2 * nb * HIDDEN + 2 * HIDDEN) + 64
        self._amax_arena = torch.zeros((need,), device=self.device, dtype=torch.float32)
        self._amax_used = 0

    def _scales(self, batch, outer):
        n = 2 * batch * outer
        o = self._amax_used
        self._amax_used = (o + n + 3) // 4 * 4
        assert self._amax_used <= self._amax_arena.numel()
        return self._amax_arena[o:o + n].view(2, batch, outer)

    def _fc8_operands(self, w8, b8):
        C, ld8 = self.C, self.ld8
        if ld8 == 2 * C:
            return w8, b8
        w8p = torch.zeros((2, ld8, HIDDEN), device=self.device, dtype=torch.float32)
        b8p = torch.zeros((2, ld8), device=self.device, dtype=torch.float32)
        w8p[:, :2 * C].copy_(w8)
        b8p[:, :2 * C].copy_(b8)
        return w8p, b8p

    def forward_backward(self, data, rois, obn_scores, labels_oh, compute_grads=True, seg=None,
                         _fuse_update=False):
        """One training pass over this GPU's images.  Returns dict of loss tensors
        (per image) and keeps what backward / stats need.  `seg` = host list of per-image row
        offsets [0, R0, R0+R1, ...] (the loader knows it); when omitted it is derived from
        rois[:,0] on the device, which costs a device->host sync per iteration."""
        C = self.C
        n_img = data.shape[0]
        if seg is None:
            seg = self.segments(rois, n_img)
        rt = rois.shape[0]
        max_seg = max(seg[i + 1] - seg[i] for i in range(n_img))
        seg_off = self._seg_to_device(seg)
        pev = getattr(self, 'phase_events', None)   # bench.py: per-stage HIP events (main stream)

        def mark(name):
            if pev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                pev.append((name, e))
        mark('start')
        conv5 = self.conv_body(data, roi_job=(rois, obn_scores, seg))
        mark('conv_body')
        x = self._roi_features(conv5, rois, obn_scores)
        del conv5
        mark('roi_pool')
        # previous iteration's all-reduce + SGD, now overlapped: joined here as a whole, or - the
        # pipelined N > 1 update - piece by piece inside the head forward
        pieces = self._join_update_for_head()
        mark('join_update')
        h6, h7, lg = self.head_forward(x, train=True, pieces=pieces)
        mark('head_fwd')
        ld8 = self.ld8
        cols = [0, C, ld8, ld8 + C]                          # fc8c, fc8d, noisy_fc8c, noisy_fc8d
        lv = [lg[:, o:o + C] for o in cols]
        ac, ad, rp, cp = ops.wsddn_outputs(lv[0], lv[1], lv[2], lv[3], seg_off)
        cw, cwn, hs, hsn = ops.entropy_gate(rois, rp[0], cp[0], labels_oh, seg_off, max_seg)
        # [class_weight | class_weight_noise] as one [2, nseg, C] view of the gate's output buffer;
        # both branches score against the same labels_oh (no stack / expand / ones kernels: the
        # loss tail is the hand-written kernels only)
        wts = cw._base[:2] if cw._base is not None and cw._base.dim() == 3 else torch.stack([cw, cwn])
        lab = labels_oh if labels_oh.is_contiguous() else labels_oh.contiguous()
        losses = ops.weighted_ce_shared(cp, lab, wts, self.is_mean)        # [2*nseg]
        out = dict(loss_cls=losses[:n_img], loss_cls_noise=losses[n_img:], cls_prob=cp[0],
                   cls_prob_noise=cp[1], class_weight=cw, class_weight_noise=cwn,
                   hatE_sum=hs, hatE_sum_norm=hsn, rois_pred=rp[0],
                   logits=torch.cat(lv, 1) if ld8 != 2 * C else lg)   # fc8c|fc8d|noisy_fc8c|noisy_fc8d
        self.step_count += 1
        mark('loss_tail')
        if not compute_grads:
            return out
        # ---- backward (loss gradient seed 1.0 per loss, blob.py:167-173)
        g = ops.weighted_ce_shared_grad(cp, lab, wts, self.is_mean, dy_const=1.0)
        dl = (torch.zeros if ld8 != 2 * C else torch.empty)((rt, 2 * ld8), device=self.device,
                                                            dtype=torch.float32)
        ops.wsddn_outputs_grad(ac, ad, rp, cp, g, seg_off, out=dl, col_offsets=cols)
        self._head_backward(x, h6, h7, dl, fuse_update=_fuse_update)
        mark('head_bwd')
        out['d_logits'] = dl if ld8 == 2 * C else torch.cat([dl[:, o:o + C] for o in cols], 1)
        return out

    def _head_backward(self, x, h6, h7, dl, fuse_update=False):
        C, rt = self.C, h6.shape[0]
        G = self.grads
        scale = 1.0 / (1.0 - self.dropout) if self.dropout > 0 else 1.0
        w7 = self.arena.span(self.params, 'fc7_w', '_[noisy]_fc7_w').view(2, HIDDEN, HIDDEN)
        w8 = self.arena.span(self.params, 'fc8c_w', 'noisy_fc8d_w').view(2, 2 * C, HIDDEN)
        gw6 = self.arena.span(G, 'fc6_w', '_[noisy]_fc6_w').view(2 * HIDDEN, self.k6)
        gb6 = self.arena.span(G, 'fc6_b', '_[noisy]_fc6_b')
        gw7 = self.arena.span(G, 'fc7_w', '_[noisy]_fc7_w').view(2, HIDDEN, HIDDEN)
        gb7 = self.arena.span(G, 'fc7_b', '_[noisy]_fc7_b')
        gw8 = self.arena.span(G, 'fc8c_w', 'noisy_fc8d_w').view(2, 2 * C, HIDDEN)
        gb8 = self.arena.span(G, 'fc8c_b', 'noisy_fc8d_b')
        ld8 = self.ld8
        pad = ld8 != 2 * C
        dlv = dl.view(rt, 2, ld8).permute(1, 0, 2)             # [2, Rt, ld8]
        h7v = h7.view(rt, 2, HIDDEN).permute(1, 0, 2)
        h6v = h6.view(rt, 2, HIDDEN).permute(1, 0, 2)
        bf = self.mfma_dtype == 'bf16'
        x3 = self.mfma_dtype == 'fp32x3'
        h2 = self.mfma_dtype == 'fp16x2'
        red = self.reducer
        # Order: the data-gradient chain first, then fc6_w's gradient (86 % of the all-reduce
        # bytes) so that its exchange starts as early as possible, the small gradients last.
        # 1. dZ7 = dL W8 and dZ6 = dZ7 W7, each gated by the ReLU / Dropout of its layer
        w8g = w8
        if pad:
            w8g = self._fc8_operands(w8, self.arena.span(self.params, 'fc8c_b', 'noisy_fc8d_b')
                                     .view(2, 2 * C))[0]
        dz7 = torch.empty_like(h7)
        dz7v = dz7.view(rt, 2, HIDDEN).permute(1, 0, 2)
        dz6 = torch.empty_like(h6)
        dz6v = dz6.view(rt, 2, HIDDEN).permute(1, 0, 2)
        planes_x = h2 and isinstance(x, ops.F16x2)
        if h2:
            # every GEMM reports the maxima its consumer's operand split needs; each of dZ7 / dZ6
            # is then read once by a split that writes all the forms in which it is multiplied
            sc7n, sc7t, sc6t = self._scales(2, rt), self._scales(2, HIDDEN), self._scales(1, 2 * HIDDEN)
            amax7 = dict(rowmax=ops.amax_words(sc7n), colmax=ops.amax_words(sc7t))
        else:
            amax7 = {}
        ops.gemm(dlv, w8g, False, False, out=dz7v, epilogue=L.EPI_GATE_POS, aux=h7v, alpha=scale,
                 **amax7)
        if h2:
            dz7n, dz7t = ops.split_f16x2_dual(dz7v, sc7n, sc7t)
            # dX = dZ W: W^T planes are kept beside the W planes.  x's planes carry per-roi scales
            # s_r: dW6 = sum_r (dZ6[r] / s_r) (x[r] s_r), so dZ6's column maxima are taken over
            # dZ6[r] / s_r (exact: powers of two)
            ops.gemm_f32_f16x2_nt(dz7n, self._wplanes['w7t'], out=dz6v,
                                  epilogue=L.EPI_GATE_POS, aux=h6v, alpha=scale,
                                  colmax=ops.amax_words(sc6t),
                                  colmax_rowmul=x.inv_scale if planes_x else None)
            del dz7n
        elif x3:
            ops.gemm_f32x3_nt(ops.split_bf16x3(dz7v), self._wplanes['w7t'], out=dz6v,
                              epilogue=L.EPI_GATE_POS, aux=h6v, alpha=scale)
        elif bf:
            ops.gemm_bf16_slab_nt(ops.to_bf16_slab(dz7v), self._wplanes['w7t'], out=dz6v,
                                  epilogue=L.EPI_GATE_POS, aux=h6v, alpha=scale)
        else:
            ops.gemm(dz7v, w7, False, False, out=dz6v, epilogue=L.EPI_GATE_POS, aux=h6v,
                     alpha=scale)
        # 2. fc6: dW = dZ6^T X in row chunks (both operands K(=rows)-contiguous through the
        # transposing split); each chunk's all-reduce starts while the next one is computed
        if h2:
            # planes [2, Rt/16, 8192, 16] of (diag(1/s_r) dZ6)^T
            dz6t = ops.split_f16x2_dual(dz6, None, sc6t.view(2, 2 * HIDDEN),
                                        rowmul=x.inv_scale if planes_x else None)[1]
            # x^T: the pooling kernel's planes are read as they are, through transposing LDS reads
            # (csrc/gemm_btr.hip); fp32 features (a caller's own roi_feat) are split transposed
            xk = planes_x
            xt = None if xk else ops.split_f16x2(x, transpose=True)    # [2, Rt/16, 25088, 16]
        elif x3:
            dz6t = ops.split_bf16x3(dz6, transpose=True)       # [3, Rt/16, 8192, 16]
            xt = ops.split_bf16x3(x, transpose=True)           # [3, Rt/16, 25088, 16]
        elif bf:
            dz6t = ops.to_bf16_slab(dz6, transpose=True)       # [Rt/16, 8192, 16]
            # [Rt/16, 25088, 16]: from the pooling kernel's own operand, or a caller's fp32 features
            xt = (ops.bf16_slab_transpose(x) if x.dtype == torch.bfloat16
                  else ops.to_bf16_slab(x, transpose=True))
        pipelined = self._pipelined()
        plan = message_plan(self.arena, 2 * HIDDEN, self.allreduce_chunks, red.active, pipelined)
        self._pipe_sent = pipelined
        # train_step() (fuse_update): each part of the piece-by-piece update is queued on the
        # update stream RIGHT BEHIND the message it waits for, from here, instead of after the
        # whole backward - fc6_w, its planes and fc6's biases are not read again once the forward
        # has run.  In queue order the update's parts then sit between the messages, so the
        # pipeline also holds where HIP puts the update stream and the collective's stream on one
        # hardware queue (two streams on one queue run in order; measured with ROCm's default
        # queue count: queued after backward, every part waited behind the whole exchange).
        eager = bool(pipelined and fuse_update and self._pipe_planes_ok())
        if eager:
            if self._upd_stream is None:
                self._upd_stream = side_stream(self.device, 'update')
            self.flush()                 # (nothing can be pending here: the head joined it)
            with torch.cuda.stream(self._upd_stream):
                self._pipe_begin()
        self._pipe_eager = eager
        if pipelined:
            # fc6's bias gradients first (the first forward piece of the next iteration needs the
            # updated biases): a column sum of dZ6, known before the weight gradient
            ops.colsum(dz6, out=gb6)
            red.reduce_async(message_slice(self.arena, G, 'fc6_b', None, self.k6))
            plan = plan[1:]
            if eager:
                with torch.cuda.stream(self._upd_stream):
                    self._pipe_enqueue_bias()
        next_piece = 0
        for kind, rows in plan[:-1]:
            r0, r1 = rows
            if h2:
                # 25088 = 98 column tiles of 256: 32 x 98 = 12.25 waves of 256 CUs.  96 column
                # tiles make 12 full waves; the last 512 columns go out as one wave of 128x128 tiles
                ncut = self._wgrad_column_cut(r1 - r0)
                if xk and fuse_update and self._can_fuse_wgrad_update():
                    self._wgrad_update_w6(dz6t, x, r0, r1, ncut)
                elif xk:
                    for c0, c1 in (((0, ncut), (ncut, self.k6)) if ncut else ((0, self.k6),)):
                        ops.gemm_f32_f16x2_nt_xk(dz6t.rows(r0, r1), x, ncols=(c0, c1),
                                                 out=gw6[r0:r1, c0:c1])
                elif ncut:
                    ops.gemm_f32_f16x2_nt(dz6t.rows(r0, r1), xt.rows(0, ncut), out=gw6[r0:r1, :ncut])
                    ops.gemm_f32_f16x2_nt(dz6t.rows(r0, r1), xt.rows(ncut, self.k6),
                                          out=gw6[r0:r1, ncut:])
                else:
                    ops.gemm_f32_f16x2_nt(dz6t.rows(r0, r1), xt, out=gw6[r0:r1])
            elif x3:
                ops.gemm_f32x3_nt(dz6t[:, :, r0:r1], xt, out=gw6[r0:r1])
            elif bf and fuse_update and self._can_fuse_wgrad_update():
                self._wgrad_update_w6_bf16(dz6t, xt, r0, r1)
            elif bf:
                ops.gemm_bf16_slab_nt(dz6t[:, r0:r1], xt, out=gw6[r0:r1])
            else:
                ops.gemm(dz6[:, r0:r1], x, True, False, out=gw6[r0:r1])
            blocks = self._shard_blocks()
            if blocks is None:
                red.reduce_async(message_slice(self.arena, G, kind, rows, self.k6))
            else:      # each owner's rows of this chunk go to that owner only
                for o, p0, p1 in owner_pieces(r0, r1, blocks):
                    red.reduce_to_owner_async(gw6[p0:p1].reshape(-1), o)
            while eager and next_piece < len(self.FWD_PIECES) and self.FWD_PIECES[next_piece][1] <= r1:
                with torch.cuda.stream(self._upd_stream):       # every chunk of this piece is out
                    self._pipe_enqueue_piece(next_piece)
                next_piece += 1
        # 3. the small gradients (under the fc6_w exchange): fc6 db; fc7 dW = dZ7^T H6, db;
        # fc8 dW = dL^T H7, db
        if not pipelined:
            ops.colsum(dz6, out=gb6)
        if h2:
            ops.gemm_f32_f16x2_nt(dz7t, self._h6t, out=gw7)
            self._h6t = None
        elif x3:
            ops.gemm_f32x3_nt(ops.split_bf16x3(dz7v, transpose=True),
                              ops.split_bf16x3(h6v, transpose=True), out=gw7)
        elif bf:
            ops.gemm_bf16_slab_nt(ops.to_bf16_slab(dz7v, transpose=True),
                                  ops.to_bf16_slab(h6v, transpose=True), out=gw7)
        else:
            ops.gemm(dz7v, h6v, True, False, out=gw7)
        ops.colsum(dz7, out=gb7)
        if pad:
            gw8p = torch.empty((2, ld8, HIDDEN), device=self.device, dtype=torch.float32)
            self._fc8_gemm(dlv, h7v, True, False, gw8p)
            gw8.copy_(gw8p[:, :2 * C])
            gb8.view(2, 2 * C).copy_(ops.colsum(dl).view(2, ld8)[:, :2 * C])
        else:
            self._fc8_gemm(dlv, h7v, True, False, gw8)
            ops.colsum(dl, out=gb8)
        red.reduce_async(message_slice(self.arena, G, *plan[-1], self.k6))
        if eager:
            self._pipe_state['grads_ready'] = torch.cuda.current_stream(self.device).record_event()
            with torch.cuda.stream(self._upd_stream):
                self._pipe_enqueue_tail()
                self._upd_event = self._upd_stream.record_event()
            self._update_pending, self._update_waiting = True, False

    def _can_fuse_wgrad_update(self):
        return (self.fuse_wgrad_update and not self.reducer.active and self.iter_size == 1
                and self.mfma_dtype in ('fp16x2', 'bf16') and self.fused_planes
                and self._sgd_regions is not None and self._wplanes is not None
                and not self._planes_dirty and self.k6 % 256 == 0)

    def _wgrad_update_w6(self, dz6t, x, r0, r1, ncut):
        """fc6_w's rows r0..r1: wgrad GEMM with the SGD update + the rows' operand planes in its
        epilogue (ops.gemm_f32_f16x2_nt_xk_sgd), on the main stream.  First chunk: the rows' maxima
        become the bounds of this update; last chunk: the conditional exact re-split."""
        n6 = 2 * HIDDEN
        sc = self._wscales.view(2, 2, n6).view(torch.int32)        # [operand][maxima | 1/scale][rows]
        maxima, bound = sc[0, 0], self._wbound[:n6]
        tag = self.sgd_iter_count + 1
        if r0 == 0:
            bound.copy_(maxima)
            maxima.zero_()
        w6 = self.arena.span(self.params, 'fc6_w', '_[noisy]_fc6_w').view(n6, self.k6)
        m6 = self.arena.span(self.momentum_buf, 'fc6_w', '_[noisy]_fc6_w').view(n6, self.k6)
        ends, lr_mult, wd = self._seg_host
        o6 = self.arena.offsets['fc6_w'][0]
        sg = next(i for i, e in enumerate(ends) if e > o6)
        assert ends[sg] >= o6 + n6 * self.k6       # one hyper-parameter run over both branches
        for c0, c1 in (((0, ncut), (ncut, self.k6)) if ncut else ((0, self.k6),)):
            ops.gemm_f32_f16x2_nt_xk_sgd(dz6t.rows(r0, r1), x, w6, m6, self.lr, lr_mult[sg], wd[sg],
                                         self.momentum, 0, self.gpu_num, self.sgd_iter_count,
                                         self._wplanes['w6'].planes, bound, maxima,
                                         self._wscales.view(2, 2, n6)[0, 1], self._wovf, tag,
                                         ncols=(c0, c1), rows=(r0, r1))
        if r1 == n6:
            ops.split_f16x2_rows_if(w6, maxima, self._wplanes['w6'], self._wovf, tag)
            # the route is decided HERE: the deferred kernel of this step must skip fc6_w whatever
            # the toggles say by the time it runs (the table to use travels with the flag)
            self._w6_updated = self._sgd_regions_rest

    def _wgrad_update_w6_bf16(self, dz6t, xt, r0, r1):
        """The bf16 plan's form of _wgrad_update_w6: fc6_w's rows r0..r1 updated (parameters,
        momentum, the rounded operand plane) in the epilogue of their weight-gradient GEMM
        (ops.gemm_bf16_slab_nt_sgd); no scales, nothing to bound or re-split."""
        n6 = 2 * HIDDEN
        w6 = self.arena.span(self.params, 'fc6_w', '_[noisy]_fc6_w').view(n6, self.k6)
        m6 = self.arena.span(self.momentum_buf, 'fc6_w', '_[noisy]_fc6_w').view(n6, self.k6)
        ends, lr_mult, wd = self._seg_host
        o6 = self.arena.offsets['fc6_w'][0]
        sg = next(i for i, e in enumerate(ends) if e > o6)
        assert ends[sg] >= o6 + n6 * self.k6       # one hyper-parameter run over both branches
        ops.gemm_bf16_slab_nt_sgd(dz6t[:, r0:r1], xt, w6, m6, self.lr, lr_mult[sg], wd[sg],
                                  self.momentum, 0, self.gpu_num, self.sgd_iter_count,
                                  self._wplanes['w6'], rows=(r0, r1))
        if r1 == n6:
            self._w6_updated = self._sgd_regions_rest

    def train_step(self, data, rois, obn_scores, labels_oh, seg=None):
        """forward_backward + sgd_step as one call.  With no gradient exchange between the two
        (one process; the reference adds its all-reduce ops only for NUM_GPUS > 1,
        optimizer_wsl.py:52-72) fc6_w's update runs in the epilogue of its wgrad GEMM: same
        arithmetic, same planes, but the gradient (0.82 GB written and read back) never exists
        and the update's traffic hides inside an MFMA-bound kernel instead of time-slicing with
        the next conv body.  `self.grads` then holds no fc6_w gradient.  With an exchange, or
        fuse_wgrad_update = False, this is exactly forward_backward(); sgd_step()."""
        out = self.forward_backward(data, rois, obn_scores, labels_oh, seg=seg, _fuse_update=True)
        self.sgd_step()
        return out

    def _wgrad_column_cut(self, rows, cus=256):
        """fc6 wgrad tile quantisation: with 256x256 tiles the [rows, k6] output is tm x tn tiles;
        if the last partial wave of `cus` tiles is less than half full, return the column where
        to cut so that the first launch is whole waves (the rest runs as small tiles), else 0."""
        tm, tn = (rows + 255) // 256, (self.k6 + 255) // 256
        tail = (tm * tn) % cus
        if tail == 0 or tail * 2 > cus or tail % tm != 0:
            return 0
        ncut = (tn - tail // tm) * 256
        return ncut if 0 < ncut < self.k6 else 0

    def wait_allreduce(self):
        self.reducer.wait()

    # -------------------------------------------------------------------- SGD
    def set_lr(self, new_lr):
        """UpdateWorkspaceLr + momentum correction (detector.py:509-559)."""
        new_lr = float(np.float32(new_lr))
        if self._lr_host == new_lr:
            return new_lr
        self.flush()                 # a pending update must see the lr it was computed under
        cur = self._lr_host
        self._lr_host = new_lr
        if cur != new_lr:
            ratio = max(new_lr / max(cur, 1e-10), cur / max(new_lr, 1e-10))
            self.lr.fill_(new_lr)
            if self.scale_momentum and cur > 1e-7 and ratio > self.scale_momentum_threshold:
                # the correction is a float32 quotient in the reference (np.float32 / np.float32,
                # detector.py:536) and the Scale op's float argument
                corr = float(np.float32(new_lr) / np.float32(cur))
                ops.unary(L.UN_SCALE, self.momentum_buf, corr, out=self.momentum_buf)
        return new_lr

    def sgd_step(self):
        """Queue this iteration's update.  By default it goes to a side stream: it waits there
        for the gradient all-reduce and runs the (HBM-bound) fused SGD kernel while the main
        stream already runs the next iteration's (MFMA-bound, parameter-free) conv body + RoIPool;
        `flush()` — called before the head touches the parameters — joins the two streams."""
        if getattr(self, '_pipe_eager', False):
            self._pipe_eager = False      # train_step: backward has queued the update already
            return
        defer = True if self.defer_update is None else self.defer_update
        if not defer:
            self._apply_update()
            return
        main = torch.cuda.current_stream(self.device)
        if self._upd_stream is None:
            # an ordinary stream: a low-priority one, or one confined to a subset of the compute
            # units, measured worse in every form (docs/history: 13.7-23 vs 13.5 ms per step)
            self._upd_stream = side_stream(self.device, 'update')
        self._grads_ready = main.record_event()
        # launched by the next conv body behind conv1_1 of every image (or by flush): started at
        # once, the SGD kernel's workgroups fill the CUs and the two conv1_1 launches at the head
        # of the dependent chains took 0.7 ms each
        self._update_waiting = True
        self._update_pending = True

    def _launch_update(self, after):
        """Queue the update on its side stream, behind the gradients and the events in `after`."""
        self._update_waiting = False
        self._upd_stream.wait_event(self._grads_ready)
        for e in after:
            self._upd_stream.wait_event(e)
        self._piece_events = None
        with torch.cuda.stream(self._upd_stream):
            self._apply_update()
            self._upd_event = self._upd_stream.record_event()

    def flush(self):
        if self._update_pending:
            if self._update_waiting:
                self._launch_update(())
            self._update_pending = False
            self._piece_events = None
            torch.cuda.current_stream(self.device).wait_event(self._upd_event)

    def _join_update_for_head(self):
        """Before the head reads the parameters: join the deferred update - entirely (-> None), or,
        when it ran piece by piece (_apply_update_pipelined), hand the pieces' events to
        head_forward, which waits for each one right before the launch that needs it."""
        if not self._update_pending:
            return None
        if self._update_waiting:
            self._launch_update(())
        if self._piece_events is None:
            self.flush()
            return None
        pieces, self._piece_events = self._piece_events, None
        return pieces

    def _finish_pieces(self, pieces):
        """The main stream joins what is left of a pipelined update (everything after fc6_w)."""
        self._update_pending = False
        torch.cuda.current_stream(self.device).wait_event(pieces['done'])

    def _pipelined(self):
        """True when this process's updates run piece by piece (NAWS.PIPELINE_UPDATE)."""
        want = self.pipeline_update
        if want is None:
            want = True
        on = bool(want and self.reducer.active and hasattr(self.reducer, 'wait_first')
                  and self.mfma_dtype == 'fp16x2'
                  and not self.sharded_update and self.iter_size == 1 and self.k6 % 256 == 0
                  and self.fused_planes and (self.defer_update is None or self.defer_update))
        if on and not getattr(self, '_pipe_warned', False):
            self._pipe_warned = True
            if self.pg is not None and self.world_size > 1:
                import torch.distributed as dist
                if dist.get_backend(self.pg) != 'gloo':
                    # per-message Work.wait() on the update stream from inside backward,
                    # SGD pieces between collectives still in flight, fc6 forward pieces gated on
                    # per-piece events - validated over gloo (2 and 8 ranks on one GPU) and with
                    # one-rank RCCL only; no multi-GPU node was ever available
                    import warnings
                    warnings.warn('NAWS.PIPELINE_UPDATE on backend %r with %d ranks has never run '
                                  'on hardware (validated over gloo and one-rank RCCL only); '
                                  'NAWS.PIPELINE_UPDATE False is the fallback: the one-launch update '
                                  'behind the whole exchange.  bench.py checks the ranks\' state '
                                  'digests after warm-up and falls back by itself on a stall'
                                  % (dist.get_backend(self.pg), self.world_size))
        return on

    def _apply_update(self):
        if self._pipe_ready():
            return self._apply_update_pipelined()      # (waits message by message itself)
        cev = getattr(self, 'comm_events', None)   # bench.py: how long this stream, with this
        if cev is not None and self.reducer.active:  # rank's gradients complete, waits for the exchange
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            self.wait_allreduce()
            c1.record()
            cev.append((c0, c1))
        else:
            self.wait_allreduce()
        if self._shard_blocks() is not None:
            return self._apply_update_sharded()
        uev = getattr(self, 'update_events', None)   # bench.py: HIP events on the update stream
        if uev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # fp16x2: the SGD kernel reports max|w| of every updated fc6_w / fc7_w row into the scale
        # blocks of their operand planes, so the re-split reads the weights once (k6 % 256: a
        # wave's 256 floats stay in one row)
        fused = (self.mfma_dtype == 'fp16x2' and self._wplanes is not None
                 and self.iter_size == 1 and self.k6 % 256 == 0)
        # (planes marked dirty - a caller wrote through blob() - carry stale maxima: that update
        # takes the exact route below)
        planes = (fused and self.fused_planes and self._sgd_regions is not None
                  and not self._planes_dirty)
        # fp32x3 / bf16: the same kernel with their plane formats (no maxima involved)
        planes_bf = (self.mfma_dtype in ('fp32x3', 'bf16') and self.fused_planes
                     and self._sgd_regions is not None and self.iter_size == 1
                     and not self._planes_dirty and self._wplanes is not None)
        rowmax = None
        rest, self._w6_updated = self._w6_updated, None
        w6_done = rest is not None
        if w6_done:
            # fc6_w (weights, momentum, planes) was updated by its wgrad GEMM in this step's
            # backward: the rest MUST take the plane-writing kernel with fc6_w's region skipped,
            # even if an A/B tool has flipped fused_planes since
            if (self._wplanes is None or self._sgd_regions is None
                    or self.mfma_dtype not in ('fp16x2', 'bf16')):
                raise RuntimeError('fc6_w was updated in its wgrad epilogue but the plane state '
                                   'it belongs to is gone (engine toggles changed mid-step?)')
            if self.mfma_dtype == 'bf16':
                planes_bf = True
            else:
                fused = planes = True
        if fused:
            maxima = self._wscales.view(2, 2, 2 * HIDDEN)[:, 0]
            part = slice(1, 2) if w6_done else slice(0, 2)      # (fc6_w's half was handled by its wgrad)
            if planes:
                self._wbound.view(2, 2 * HIDDEN)[part].copy_(maxima.view(torch.int32)[part])
            maxima[part].zero_()
            rowmax = self._wscales.view(torch.int32)
        if planes:
            tag = self.sgd_iter_count + 1
            self._wplanes['w7t'].scales[0].zero_()
            ops.acm_sgd_update_f16x2(self.grads, self.momentum_buf, self.lr, self.params,
                                     self.seg_end, self.seg_lr_mult, self.seg_wd, self.momentum, 0,
                                     self.gpu_num, self.sgd_iter_count,
                                     rest if w6_done else self._sgd_regions,
                                     self._wovf, tag)
        elif planes_bf:
            ops.acm_sgd_update_planes(self.grads, self.momentum_buf, self.lr, self.params,
                                      self.seg_end, self.seg_lr_mult, self.seg_wd, self.momentum, 0,
                                      self.gpu_num, self.sgd_iter_count,
                                      rest if w6_done else self._sgd_regions)
        else:
            ops.acm_sgd_update(self.grads, self.momentum_buf, self.lr, self.params, self.acmgrad,
                               self.seg_end, self.seg_lr_mult, self.seg_wd, self.momentum, 0,
                               self.iter_size, self.gpu_num, self.sgd_iter_count, rowmax=rowmax,
                               rm_table=self._rm_table if fused else None)
        if uev is not None:
            e1.record()
            uev.append((e0, e1))
        self.sgd_iter_count += 1
        if planes:
            # the planes are written; should a row have outgrown twice its old maximum, these two
            # launches redo them from the exact maxima (otherwise their workgroups leave at once)
            w6, w7 = self._weight_views()
            wp = self._wplanes
            sc = self._wscales.view(2, 2, 2 * HIDDEN).view(torch.int32)
            if not w6_done:
                ops.split_f16x2_rows_if(w6, sc[0, 0], wp['w6'], self._wovf, tag)
            ops.split_f16x2_rows_if(w7, sc[1, 0], wp['w7'], self._wovf, tag)
            # fc7_w^T from the column maxima the SGD kernel has just reported: one pass
            # (was: a maxima pass + a transposing split, 92 + 116 us)
            ops.split_f16x2_dual(w7, None, wp['w7t'].scales, out_t=wp['w7t'])
            self._planes_dirty = False
        elif fused:                            # same stream as the update: hidden with it
            w6, w7 = self._weight_views()
            wp = self._wplanes
            q6 = ops.split_f16x2_dual(w6, wp['w6'].scales, None, out_n=wp['w6'])[0]
            q7 = ops.split_f16x2_dual(w7, wp['w7'].scales, None, out_n=wp['w7'])[0]
            assert q6 is wp['w6'] and q7 is wp['w7']
            ops.split_f16x2(w7, transpose=True, out=wp['w7t'])
            self._planes_dirty = False
        elif planes_bf:
            cv = ops.split_bf16x3 if self.mfma_dtype == 'fp32x3' else ops.to_bf16_slab
            cv(self._weight_views()[1], transpose=True, out=self._wplanes['w7t'])
            self._planes_dirty = False
        elif self.mfma_dtype != 'fp32' and self._wplanes is not None:
            self._refresh_weight_planes()

    # ------------------------------------------------- NAWS.PIPELINE_UPDATE
    FWD_PIECES = ((0, HIDDEN), (HIDDEN, 2 * HIDDEN))      # one branch each: 16 x 16 tiles of 256 x 256
                                                          # at R = 4000 = one round of the 256 CUs

    def _pipe_ready(self):
        """The messages of this step went out in the pipelined order AND the plane-writing kernel
        can run on them (planes valid, one hyper-parameter run per weight matrix); otherwise the
        update waits for everything and takes the one-launch route - any message order is fine
        for that."""
        return (getattr(self, '_pipe_sent', False) and self._pipe_planes_ok()
                and self._upd_stream is not None
                and torch.cuda.current_stream(self.device) == self._upd_stream)

    def _pipe_planes_ok(self):
        return (self._wplanes is not None and self._sgd_regions is not None
                and not self._planes_dirty)

    def _seg_tables(self, start, count):
        """(seg_end, seg_lr_mult, seg_wd) device tensors of the arena range [start, start + count),
        ends relative to `start` (the SGD kernels take a base pointer + per-run hyper-parameters)."""
        ends, lr_mult, wd = self._seg_host
        e2, l2, w2 = [], [], []
        for e, l, w in zip(ends, lr_mult, wd):
            if e <= start:
                continue
            e2.append(min(e, start + count) - start)
            l2.append(l)
            w2.append(w)
            if e >= start + count:
                break
        return (torch.tensor(e2, dtype=torch.int64, device=self.device),
                torch.tensor(l2, dtype=torch.float32, device=self.device),
                torch.tensor(w2, dtype=torch.float32, device=self.device))

    def _pipe_tables(self):
        if self._pipe is None:
            n6 = 2 * HIDDEN
            o6, ob, o7 = (self.arena.offsets[k][0] for k in ('fc6_w', 'fc6_b', 'fc7_w'))
            sc = self._wscales.view(2, 2, n6)                # [operand][maxima | 1/scale][rows]
            p6, p7 = self._wplanes['w6'], self._wplanes['w7']
            cm7 = self._wplanes['w7t'].scales[0].view(torch.int32)
            pieces = []
            for r0, r1 in self.FWD_PIECES:
                regs = ops.SgdPlaneRegions([(0, r1 - r0, self.k6, n6, (p6.planes, r0),
                                             self._wbound[r0:r1], sc[0, 0].view(torch.int32)[r0:r1],
                                             sc[0, 1][r0:r1])])
                pieces.append(dict(rows=(r0, r1), start=o6 + r0 * self.k6, count=(r1 - r0) * self.k6,
                                   regions=regs, seg=self._seg_tables(o6 + r0 * self.k6,
                                                                      (r1 - r0) * self.k6)))
            tail = ops.SgdPlaneRegions([(0, n6, HIDDEN, HIDDEN, p7.planes, self._wbound[n6:],
                                         sc[1, 0].view(torch.int32), sc[1, 1], cm7)])
            self._pipe = dict(pieces=pieces, bias=dict(start=ob, count=o7 - ob,
                                                       seg=self._seg_tables(ob, o7 - ob)),
                              tail=dict(start=o7, count=self.arena.total - o7, regions=tail,
                                        seg=self._seg_tables(o7, self.arena.total - o7)),
                              ovf=torch.zeros((len(self.FWD_PIECES) + 1,), device=self.device,
                                              dtype=torch.int32))
        return self._pipe

    def _apply_update_pipelined(self):
        """The deferred update of an N > 1 step, piece by piece (update stream).  Messages arrive
        in the order backward handed them over: [fc6 biases][fc6_w row chunks ...][the rest].
          1. fc6's biases (element-wise kernel on their 8192 floats);
          2. per forward piece (FWD_PIECES: fc6_w rows of one branch): wait for the chunks that
             cover it, run the plane-writing SGD kernel on exactly those rows (arena slice, row
             block of the operand planes), queue the conditional exact re-split of those rows,
             record the piece's event - the next iteration's fc6 forward launches that piece
             behind it (head_forward) while the later messages are still on the links;
          3. the rest (fc7_w with planes + column maxima, fc8, biases), fc7_w's transposed planes,
             the final event.
        Per element the arithmetic is the one-launch route's (same kernels on slices: parameters,
        momentum, planes and scales bit-identical, tests/test_gpu_two_ranks.py); the one difference:
        the "a row outgrew twice its previous maximum" word is kept PER PIECE, so the exact
        re-split that answers it covers that piece's rows instead of all of fc6_w / fc7_w - the
        other rows keep their (equally valid) bound-derived scales.

        This is the form sgd_step() queues after a plain forward_backward().  train_step() queues
        the same three parts EARLIER, from inside backward, each right behind the message it
        waits for (_pipe_enqueue_*): fc6_w is not read again once its forward has run, so its rows
        may be updated while the rest of backward is still computing."""
        self._pipe_begin()
        self._pipe_enqueue_bias()
        for i in range(len(self.FWD_PIECES)):
            self._pipe_enqueue_piece(i)
        self._pipe_enqueue_tail()

    def _pipe_slice(self, flat, t):
        return flat[t['start']:t['start'] + t['count']]

    def _pipe_begin(self):
        """Start of one step's piece-by-piece update (current stream = the update stream)."""
        self._pipe_tables()
        self._pipe_state = dict(arrived=0, events=[], tag=self.sgd_iter_count + 1, grads_ready=None)

    def _pipe_enqueue_bias(self):
        tb, b = self._pipe, self._pipe['bias']
        self.reducer.wait_first(1)
        ops.acm_sgd_update(self._pipe_slice(self.grads, b), self._pipe_slice(self.momentum_buf, b),
                           self.lr, self._pipe_slice(self.params, b), None, b['seg'][0], b['seg'][1],
                           b['seg'][2], self.momentum, 0, 1, self.gpu_num, self.sgd_iter_count)

    def _pipe_chunks_below(self, row):
        """How many fc6_w messages start below `row` (= must have arrived before rows < row are
        complete)."""
        return sum(1 for kind, rows in message_plan(self.arena, 2 * HIDDEN, self.allreduce_chunks, True, True)
                   if kind == 'fc6_w' and rows[0] < row)

    def _pipe_enqueue_piece(self, i):
        n6 = 2 * HIDDEN
        st, pc = self._pipe_state, self._pipe['pieces'][i]
        r0, r1 = pc['rows']
        need = self._pipe_chunks_below(r1)
        self.reducer.wait_first(need - st['arrived'])
        st['arrived'] = need
        sc = self._wscales.view(2, 2, n6)
        max6 = sc[0, 0].view(torch.int32)
        self._wbound[r0:r1].copy_(max6[r0:r1])
        max6[r0:r1].zero_()
        ovf = self._pipe['ovf'][i:i + 1]
        w6, _w7 = self._weight_views()
        ops.acm_sgd_update_f16x2(self._pipe_slice(self.grads, pc), self._pipe_slice(self.momentum_buf, pc),
                                 self.lr, self._pipe_slice(self.params, pc), pc['seg'][0], pc['seg'][1],
                                 pc['seg'][2], self.momentum, 0, self.gpu_num, self.sgd_iter_count,
                                 pc['regions'], ovf, st['tag'])
        ops.split_f16x2_row_range_if(w6, max6, self._wplanes['w6'], r0, r1, cond=ovf, cond_value=st['tag'])
        st['events'].append((r0, r1, self._upd_stream.record_event()))

    def _pipe_enqueue_tail(self):
        n6 = 2 * HIDDEN
        st, t, wp = self._pipe_state, self._pipe['tail'], self._wplanes
        # bench.py (comm_events): how long this stream, with this rank's gradients complete, still
        # waits for the exchange (queued from inside backward: `grads_ready` = the main stream's
        # position behind the last gradient kernel)
        cev = getattr(self, 'comm_events', None)
        if cev is not None:
            if st['grads_ready'] is not None:
                self._upd_stream.wait_event(st['grads_ready'])
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
        self.reducer.wait()
        if cev is not None:
            c1.record()
            cev.append((c0, c1))
        uev = getattr(self, 'update_events', None)
        if uev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        sc = self._wscales.view(2, 2, n6)
        max7 = sc[1, 0].view(torch.int32)
        self._wbound[n6:].copy_(max7)
        max7.zero_()
        wp['w7t'].scales[0].zero_()
        ovf = self._pipe['ovf'][-1:]
        _w6, w7 = self._weight_views()
        ops.acm_sgd_update_f16x2(self._pipe_slice(self.grads, t), self._pipe_slice(self.momentum_buf, t),
                                 self.lr, self._pipe_slice(self.params, t), t['seg'][0], t['seg'][1],
                                 t['seg'][2], self.momentum, 0, self.gpu_num, self.sgd_iter_count,
                                 t['regions'], ovf, st['tag'])
        if uev is not None:
            e1.record()
            uev.append((e0, e1))
        self.sgd_iter_count += 1
        ops.split_f16x2_rows_if(w7, max7, wp['w7'], ovf, st['tag'])
        ops.split_f16x2_dual(w7, None, wp['w7t'].scales, out_t=wp['w7t'])
        self._planes_dirty = False
        self._piece_events = dict(fc6=st['events'], done=self._upd_stream.record_event())
        self._pipe_state = None

    # ------------------------------------------------- NAWS.SHARDED_UPDATE
    def _shard_blocks(self):
        """The owners' row blocks of fc6_w when the sharded update is in force, else None."""
        if not (self.sharded_update and self.reducer.active):
            return None
        if self._shard is None:
            if self.mfma_dtype != 'fp16x2' or self.iter_size != 1 or self.k6 % 256 != 0:
                raise NotImplementedError('NAWS.SHARDED_UPDATE needs the fp16x2 plan, ITER_SIZE 1 '
                                          'and a 256-multiple fc6 input')
            blocks = owner_blocks(2 * HIDDEN, self.reducer.world_size)
            if blocks is None:
                raise NotImplementedError('NAWS.SHARDED_UPDATE: 8192 rows do not divide into '
                                          '32-row multiples over %d ranks' % self.reducer.world_size)
            self._shard = dict(blocks=blocks)
            if self.pg is not None:
                import warnings
                import torch.distributed as dist
                if dist.get_backend(self.pg) != 'gloo':
                    # the RCCL legs of this route (per-owner dist.reduce, in-place
                    # all_gather_into_tensor on fp32 and int32 buffers, issued from the update
                    # stream) have never executed - no multi-GPU node was available; only the gloo
                    # emulation (all_reduce / broadcast) has
                    warnings.warn('NAWS.SHARDED_UPDATE on backend %r has never run on hardware '
                                  '(validated over gloo only); check the first steps against the '
                                  'all-reduce route' % dist.get_backend(self.pg))
        return self._shard['blocks']

    def _shard_tables(self):
        sh = self._shard
        if 'regions' not in sh:
            if self._wplanes is None or self._sgd_regions is None:
                raise RuntimeError('the sharded update needs the plane-writing SGD kernel '
                                   '(one hyper-parameter run over fc6_w / fc7_w)')
            n6 = 2 * HIDDEN
            b0, b1 = sh['blocks'][self.rank]
            nb = b1 - b0
            o6, o7 = self.arena.offsets['fc6_w'][0], self.arena.offsets['fc7_w'][0]
            sc = self._wscales.view(2, 2, n6)                # [operand][maxima | 1/scale][rows]
            p6, p7 = self._wplanes['w6'], self._wplanes['w7']
            cm7 = self._wplanes['w7t'].scales[0].view(torch.int32)
            regs = []
            if b0 > 0:
                regs.append((o6, b0, self.k6, 32, None, None, None, None))
            regs.append((o6 + b0 * self.k6, nb, self.k6, n6,
                         p6.planes if nb == n6 else (p6.planes, b0), self._wbound[b0:b1],
                         sc[0, 0].view(torch.int32)[b0:b1], sc[0, 1][b0:b1]))
            if b1 < n6:
                regs.append((o6 + b1 * self.k6, n6 - b1, self.k6, 32, None, None, None, None))
            regs.append((o7, n6, HIDDEN, HIDDEN, p7.planes, self._wbound[n6:],
                         sc[1, 0].view(torch.int32), sc[1, 1], cm7))
            sh['regions'] = ops.SgdPlaneRegions(regs)
            # what travels back beside the fp32 rows: per owner [row maxima | 1/scale | overflow word]
            sh['meta'] = torch.zeros((len(sh['blocks']), 2 * nb + 4), device=self.device,
                                     dtype=torch.int32)
        return sh['regions'], sh['meta']

    def _apply_update_sharded(self):
        """The deferred update of a rank under NAWS.SHARDED_UPDATE (update stream):
          1. wait for the reduces of this rank's fc6_w rows and for the small all-reduce;
          2. the plane-writing SGD kernel over [this rank's fc6_w rows | fc7_w | fc8 | biases] - the
             other owners' fc6_w rows are skip regions (neither read nor written);
          3. all-gather, in place: the updated fp32 rows (822 MB / N per rank) and, per owner, the
             rows' exact maxima, the 1/scale the owner's planes carry, and its overflow word;
          4. one split of all 8192 rows with exactly the scales the all-reduce route would hold:
             the owners' bound-derived ones, or - if ANY row of any owner outgrew its bound, which
             is what the single overflow word of the all-reduce route records - the exact maxima.
        fc7_w / fc8 / biases: as in the all-reduce route (every rank updates them)."""
        n6 = 2 * HIDDEN
        blocks = self._shard['blocks']
        b0, b1 = blocks[self.rank]
        nb = b1 - b0
        red = self.reducer
        self.wait_allreduce()
        if self._planes_dirty or self._wplanes is None:
            raise RuntimeError('sharded update: the operand planes are stale (blob() written '
                               'through between backward and update?)')
        regions, meta = self._shard_tables()
        sc = self._wscales.view(2, 2, n6)
        max6, max7 = sc[0, 0].view(torch.int32), sc[1, 0].view(torch.int32)
        tag = self.sgd_iter_count + 1
        self._wbound[b0:b1].copy_(max6[b0:b1])
        max6[b0:b1].zero_()
        self._wbound[n6:].copy_(max7)
        max7.zero_()
        self._wplanes['w7t'].scales[0].zero_()
        uev = getattr(self, 'update_events', None)
        if uev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ops.acm_sgd_update_f16x2(self.grads, self.momentum_buf, self.lr, self.params, self.seg_end,
                                 self.seg_lr_mult, self.seg_wd, self.momentum, 0, self.gpu_num,
                                 self.sgd_iter_count, regions, self._wovf, tag)
        if uev is not None:
            e1.record()
            uev.append((e0, e1))
        self.sgd_iter_count += 1
        self._mom_synced = len(blocks) == 1
        # every owner's block starts from this rank's own view of those rows (what the previous
        # gather left: right for the rows this rank does not own only until the gather below
        # overwrites them - and what an exchange that moves no data, bench.py's projection, keeps)
        meta[:, :nb].copy_(max6.view(len(blocks), nb))
        meta[:, nb:2 * nb].copy_(sc[0, 1].view(torch.int32).view(len(blocks), nb))
        meta[:, 2 * nb].zero_()
        meta[self.rank, 2 * nb:2 * nb + 1].copy_(self._wovf)
        w6, w7 = self._weight_views()
        red.gather_blocks_async(w6.reshape(-1), self.rank)
        red.gather_blocks_async(meta.view(-1), self.rank)
        red.wait()
        raised = (meta[:, 2 * nb] == tag).any()
        rowmax = meta[:, :nb].reshape(n6)
        bound_scale = (meta[:, nb:2 * nb].reshape(n6).view(torch.float32) * 16384.0).view(torch.int32)
        max6.copy_(rowmax)
        eff = torch.where(raised, rowmax, bound_scale)     # 1/scale = 2^(e-14)  ->  a maximum 2^e
        self._wovf.copy_(torch.where(raised, torch.full_like(self._wovf, tag), self._wovf))
        wp = self._wplanes
        # the rows this rank does NOT own: planes from the owners' scales (always); its own rows
        # already carry the planes its SGD kernel wrote with exactly those scales - they are redone
        # (from the exact maxima) only if some owner raised the overflow word (the full
        # re-split gave back most of what the owner-only update saves)
        if b0 > 0:
            ops.split_f16x2_row_range_if(w6, eff, wp['w6'], 0, b0)
        if b1 < n6:
            ops.split_f16x2_row_range_if(w6, eff, wp['w6'], b1, n6)
        ops.split_f16x2_row_range_if(w6, eff, wp['w6'], b0, b1, cond=self._wovf, cond_value=tag)
        ops.split_f16x2_rows_if(w7, max7, wp['w7'], self._wovf, tag)
        ops.split_f16x2_dual(w7, None, wp['w7t'].scales, out_t=wp['w7t'])
        self._planes_dirty = False

    def gather_sharded_state(self):
        """COLLECTIVE (every rank): bring the owners' momentum rows of fc6_w to all ranks so that
        any rank can write a complete checkpoint.  The fp32 weights are complete on every rank
        after each update; the momentum of a row lives with its owner only.  No-op otherwise."""
        if self._shard_blocks() is None or self._mom_synced:
            return
        self.flush()
        m6 = self.arena.span(self.momentum_buf, 'fc6_w', '_[noisy]_fc6_w')
        self.reducer.gather_blocks_async(m6, self.rank)
        self.reducer.wait()
        self._mom_synced = True

    # -------------------------------------------------------------- inference
    def infer(self, data, rois, obn_scores, seg=None):
        """Test-mode forward: cls_prob [R, C+1] = Concat(rois_pred[:, :1], rois_pred)
        (wsl_heads.py:58-67); no dropout; only the clean branch is fetched (test_wsl.py:151)."""
        if rois.shape[0] == 0:
            raise ValueError('infer: no proposals (rois is empty)')
        self.flush()
        n_img = data.shape[0]
        if seg is None:
            seg = self.segments(rois, n_img)
        seg_off = self._seg_to_device(seg)
        conv5 = self.conv_body(data, roi_job=(rois, obn_scores, seg))
        x = self._roi_features(conv5, rois, obn_scores)
        _h6, _h7, lg = self.head_forward(x, train=False, both_branches=False)
        C = self.C
        _ac, _ad, rp, _cp = ops.wsddn_outputs(lg[:, :C], lg[:, C:2 * C], None, None, seg_off)
        return torch.cat([rp[0][:, :1], rp[0]], dim=1)

    # ------------------------------------------------------------------ Stat
    def stat_update(self, out, labels_oh, display, printer=print):
        """The six Stat ops of webly_heads.py:400-440 (image 0 of this process)."""
        C = self.C
        if self.stat_state is None:
            self.stat_state = dict(AI=torch.zeros((6, C), device=self.device),
                                   AL=torch.zeros((6, C), device=self.device), it=0, init=True)
        st = self.stat_state
        lab = labels_oh[0].contiguous()
        bg = ops.unary(L.UN_SCALE, lab, -1.0)
        bg = ops.binary(L.BIN_ADD, bg.view(1, C), torch.ones((1, 1), device=self.device)).view(C)
        pairs = [('class_weight      ', out['class_weight'][0], bg),
                 ('class_weight_noise', out['class_weight_noise'][0], bg),
                 ('hatE_sum bg       ', out['hatE_sum'][0], bg),
                 ('hatE_sum fg       ', out['hatE_sum'][0], lab),
                 ('hatE_sum_norm bg  ', out['hatE_sum_norm'][0], bg),
                 ('hatE_sum_norm fg  ', out['hatE_sum_norm'][0], lab)]
        for k, (_p, i_, l_) in enumerate(pairs):
            ops.stat_accumulate(i_.contiguous(), l_, st['AI'][k], st['AL'][k], st['init'])
        st['init'] = False
        st['it'] += 1
        if st['it'] % display == 0 or st['it'] == 1:
            ai, al = st['AI'].cpu().numpy(), st['AL'].cpu().numpy()
            with np.errstate(divide='ignore', invalid='ignore'):
                ratio = ai / al
            for k, (p, _i, _l) in enumerate(pairs):
                printer('\t' + p + ' Stat #iter_: ' + str(st['it']) + ''.join(
                    ' %.2f' % v for v in ratio[k]))
            st['init'] = True
