"""World size 2 on ONE GPU: two processes share cuda:0 and exchange gradients over gloo (RCCL
refuses two ranks on one device, and the pool's boxes have one GPU).  Everything of the N > 1
path except the transport runs on hardware: the fc6_w wgrad in row chunks with one collective
per chunk, the deferred update waiting for the exchange on its side stream under the next conv
body, gpu_num = world x images in the SGD scale.  Reference: detectron/modeling/
optimizer_wsl.py:52-72 (one all-reduce per gradient blob, then the update on every GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, B, STEPS, LR = 20, 2, 3, 1e-4


def _batches(world=2):
    """world x B images' worth of inputs: rank r takes images B r .. B r + B - 1 (make_minibatch
    numbers the images of a batch from 0, so each rank's share is built on its own)."""
    from detectron.datasets import synthetic
    roidb = synthetic.make_roidb(world * B, 48, C, 96, 128, seed=5)
    halves = [synthetic.make_minibatch(roidb[r * B:(r + 1) * B], C) for r in range(world)]
    return roidb, halves


def _engine(dev, gpu_num, pg=None, world=1, sharded=False, chunks=4, pipeline=None, dropout=0.0):
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    eng = WsddnEngine(C + 1, dev, dropout=dropout, gpu_num=gpu_num, seed=5, process_group=pg,
                      world_size=world, allreduce_chunks=chunks, sharded_update=sharded,
                      pipeline_update=pipeline)
    blobs = synthetic.init_blobs(C, seed=5)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    eng.set_lr(LR)
    return eng


def _run(eng, mb, dev, train_step=False):
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    n = int(mb['data'].shape[0])
    seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=n)).tolist()
    losses = []
    for _ in range(STEPS):
        if train_step:     # (the training loop's call: the update's parts are queued from inside backward)
            out = eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
        else:
            out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            eng.sgd_step()
        losses.append(out['loss_cls'].cpu().numpy().copy())
    eng.flush()
    torch.cuda.synchronize()
    return np.stack(losses)


def _worker(rank, world, port, outdir, sharded=False, pipeline=None, dropout=0.0, tag=None,
            train_step=False):
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    import faulthandler
    # a rank stuck in a collective says where, then leaves (the parent's limit is 300 s)
    faulthandler.dump_traceback_later(float(os.environ.get('NAWS_RANK_LIMIT', '240')), exit=True)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _roidb, halves = _batches(world)
    # world 2: four explicit chunks; above: the engine's own auto choice ("2 above 2 ranks")
    eng = _engine(dev, world * B, dist.group.WORLD, world, sharded=sharded,
                  chunks=4 if world == 2 else 0, pipeline=pipeline, dropout=dropout)
    assert eng.reducer.active and (eng._shard_blocks() is not None) == sharded
    # the update runs piece by piece by default whenever there is an exchange (not when sharded)
    assert eng._pipelined() == (not sharded and pipeline is not False)
    assert eng.gpu_num == world * B
    if world > 2:
        assert eng.allreduce_chunks == 2
        if sharded:
            rows = 8192 // world
            assert eng._shard_blocks() == [(r * rows, (r + 1) * rows) for r in range(world)]
    eng.reducer.log = log = []
    losses = _run(eng, halves[rank], dev, train_step=train_step)
    if rank == 0:
        import json
        with open(os.path.join(outdir, 'messages%s.json' % (tag if tag is not None else
                                                             ('s' if sharded else ''))), 'w') as f:
            json.dump(log, f)
    tag = tag if tag is not None else ('s' if sharded else '')
    if eng._pipelined():
        assert log[0][0] == 'all_reduce' and log[0][1] == 8192          # fc6's biases travel first
    if sharded:
        # momentum rows live with their owner until a checkpoint gathers them
        try:
            eng.export_blobs()
            raise AssertionError('export_blobs must refuse before gather_sharded_state')
        except RuntimeError:
            pass
        eng.gather_sharded_state()
        eng.export_blobs()
    wp = eng._wplanes
    arrays = dict(params=eng.params.cpu().numpy(), mom=eng.momentum_buf.cpu().numpy(),
                  planes=wp['w6'].planes.view(torch.int16).cpu().numpy(),
                  scales=wp['w6'].inv_scale.cpu().numpy())
    if world == 2 or rank == 0:
        for k, v in arrays.items():
            np.save(os.path.join(outdir, '%s%s%d.npy' % (k, tag, rank)), v)
    # (world > 2: 2.7 GB per rank - the other ranks leave a digest of every array instead)
    import hashlib
    import json
    with open(os.path.join(outdir, 'digest%s%d.json' % (tag, rank)), 'w') as f:
        json.dump({k: hashlib.blake2b(np.ascontiguousarray(v).view(np.uint8).reshape(-1),
                                      digest_size=16).hexdigest() for k, v in arrays.items()}, f)
    np.save(os.path.join(outdir, 'losses%s%d.npy' % (tag, rank)), losses)
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(procs, limit=300.0):
    """Start the ranks and wait; a rank that is still alive at the limit is killed (a rank stuck in a
    collective would otherwise keep the test session from ever exiting) and the test fails."""
    import time
    for p in procs:
        p.start()
    t0 = time.time()
    for p in procs:
        p.join(timeout=max(1.0, limit - (time.time() - t0)))
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.kill()
        p.join(timeout=30)
    assert not stuck, 'ranks did not finish within %.0f s' % limit
    for p in procs:
        assert p.exitcode == 0


def test_two_ranks_on_one_gpu_equal_one_rank_with_all_four_images(dev, tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    _run_ranks(procs)
    p0, p1 = (np.load(str(tmp_path / ('params%d.npy' % r))) for r in range(2))
    assert np.array_equal(p0, p1)          # same sums in the same order on both ranks
    # one rank, the four images as one batch: the same per-image losses, and the same update up
    # to the order in which the gradient sums over proposals are taken
    from detectron.datasets import synthetic
    roidb, _halves = _batches()
    eng = _engine(dev, 2 * B)
    ref_losses = _run(eng, synthetic.make_minibatch(roidb, C), dev)
    got_losses = np.concatenate([np.load(str(tmp_path / ('losses%d.npy' % r))) for r in range(2)], 1)
    ref = eng.params.cpu().numpy()
    assert np.allclose(got_losses[0], ref_losses[0], rtol=1e-6, atol=0)      # before any update
    assert np.allclose(got_losses, ref_losses, rtol=2e-4, atol=1e-6)
    step = np.abs(ref - synthetic_flat(eng)).max()
    assert step > 0
    assert np.abs(p0 - ref).max() <= 1e-3 * step + 1e-9, (np.abs(p0 - ref).max(), step)


def synthetic_flat(eng):
    """The initial parameters, in arena order."""
    from detectron.datasets import synthetic
    blobs = synthetic.init_blobs(C, seed=5)
    out = np.empty((eng.arena.total,), np.float32)
    for name, _shape in eng.arena.specs:
        off, n, _s = eng.arena.offsets[name]
        out[off:off + n] = np.asarray(blobs[name], np.float32).reshape(-1)
    return out


def test_sharded_update_two_ranks_bit_identical_to_the_allreduce_route(dev, tmp_path):
    """NAWS.SHARDED_UPDATE on hardware (two ranks on one GPU over gloo): after three training
    steps the parameters, the momentum (once gathered), fc6_w's operand planes and their scales
    are bit-identical to the all-reduce route's, on both ranks."""
    ctx = mp.get_context('spawn')
    for sharded in (False, True):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), sharded)) for r in range(2)]
        _run_ranks(procs)
    for what in ('params', 'mom', 'planes', 'scales', 'losses'):
        ref = np.load(str(tmp_path / ('%s0.npy' % what)))
        for r in range(2):
            got = np.load(str(tmp_path / ('%ss%d.npy' % (what, r))))
            if what == 'losses':
                want = np.load(str(tmp_path / ('%s%d.npy' % (what, r))))
                assert np.array_equal(got, want), (what, r)
            else:
                assert np.array_equal(got, ref), (what, r)


# ------------------------------------------------------------------------------------------
# configs[2]'s N-rank schedule (N ranks x 2 images, gpu_num = 2 N) on the one GPU of the box.  The
# box's process guard allows six processes on the GPU at once - this session + at most five ranks -,
# so the hardware run is world 4 (rounds 4-5 ran world 8 here before the guard existed; the world-8
# message plan and owner blocks are covered on CPU: tests/test_distributed_cpu.py, test_bench_launcher.py)
MANY = 4
# ------------------------------------------------------------------------------------------
def _spawn(world, outdir, sharded):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, outdir, sharded)) for r in range(world)]
    _run_ranks(procs, limit=420.0)


def _sequential_reference(dev, world):
    """ONE process playing all `world` ranks in turn: every step, each rank's 2 images go through
    forward_backward on the SAME parameters, the `world` gradient arenas are summed (float64 on
    the device, rounded once to fp32), and one update with gpu_num = world x B follows - the
    data-parallel step with the exchange's summation order taken out.  The world-rank job differs
    from it by the order of the fp32 additions inside gloo's ring (at most a few ulps of a
    gradient sum), never by schedule, ownership or scale."""
    from detectron.datasets import synthetic
    _roidb, shares = _batches(world)
    eng = _engine(dev, world * B)
    eng.defer_update = False
    ts = []
    for mb in shares:
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)).tolist()
        ts.append((t, seg))
    losses = []
    for _ in range(STEPS):
        acc = torch.zeros_like(eng.grads, dtype=torch.float64)
        per = []
        for t, seg in ts:
            out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            acc += eng.grads.double()
            per.append(out['loss_cls'].cpu().numpy().copy())
        eng.grads.copy_(acc.float())
        eng.sgd_step()
        losses.append(np.concatenate(per))
    eng.flush()
    torch.cuda.synchronize()
    return eng, np.stack(losses)


def test_many_ranks_on_one_gpu_allreduce_and_sharded_routes(dev, tmp_path):
    """configs[2]'s schedule above two ranks: MANY processes share cuda:0 and
    exchange over gloo, 2 images each, gpu_num = 2 MANY, the engine's own auto chunk count (2),
    perm-style image shards, owner blocks of 8192 / MANY fc6_w rows on the sharded route.
      * every rank ends with bit-identical parameters (both routes);
      * the sharded route is bit-identical to the all-reduce route: parameters, momentum (once
        gathered), fc6_w's operand planes and their scales, on every rank;
      * against ONE process that plays the ranks in turn and sums their gradients itself
        (gpu_num = 2 MANY): first-step losses equal, parameters after 3 steps within the fp32
        summation-order slack of the exchange (1e-3 of the total parameter movement);
      * against one process fed all the images as ONE batch (gpu_num = 2 MANY): the same within the
        order in which the proposals of a batch are summed.
    Bit-identity between N ranks and one process is not a property of the algorithm (nor of the
    reference: NCCL's ring adds the ranks' fp32 gradients in a rank-rotated order per chunk,
    optimizer_wsl.py:52-72), so the single-process comparisons carry a stated tolerance."""
    import json
    world = MANY
    for sharded in (False, True):
        _spawn(world, str(tmp_path), sharded)
    ref = {w: np.load(str(tmp_path / ('%s0.npy' % w))) for w in ('params', 'mom', 'planes', 'scales')}
    dig = {(t, r): json.load(open(str(tmp_path / ('digest%s%d.json' % (t, r)))))
           for t in ('', 's') for r in range(world)}
    for r in range(1, world):          # (momentum included: every rank updates every row here)
        assert dig[('', r)] == dig[('', 0)], r
    # the sharded route: every rank holds the SAME parameters, momentum (once gathered), operand
    # planes and scales - the owners' rows and scale words reached everybody - bit for bit
    sref = {w: np.load(str(tmp_path / ('%ss0.npy' % w))) for w in ('params', 'mom', 'planes', 'scales')}
    for r in range(1, world):
        assert dig[('s', r)] == dig[('s', 0)], r
    # ... and against the all-reduce route: the first step's losses are equal, the parameters
    # agree within the summation-order slack.  (At world 2 the two routes are bit-identical -
    # a + b = b + a - and the two-rank test asserts that; with more addends the ring's order of
    # additions depends on where an element sits in its message, and the two routes cut
    # fc6_w's gradient into different messages: 2 row chunks vs one piece per owner.)
    for r in range(world):
        ls, la = (np.load(str(tmp_path / ('losses%s%d.npy' % (t, r)))) for t in ('s', ''))
        assert np.array_equal(ls[0], la[0]), r
        assert np.allclose(ls, la, rtol=1e-5, atol=1e-7), r
    # the message schedule rank 0 handed to the exchange, per step: 2 fc6_w row chunks of
    # 4096 x 25088 floats then the small gradients (all-reduce route); per-owner pieces of the
    # same chunks + small + the two gathers (sharded route)
    msgs = json.load(open(str(tmp_path / 'messages.json')))
    per_step = len(msgs) // STEPS
    # (the pipelined order: fc6's biases, 2 fc6_w row chunks, the rest)
    assert per_step == 4 and [m[1] for m in msgs[:4]] == [8192, 4096 * 25088, 4096 * 25088, msgs[3][1]]
    assert msgs[3][1] == ref['params'].size - 8192 * 25088 - 8192
    smsgs = json.load(open(str(tmp_path / 'messagess.json')))
    kinds = [m[0] for m in smsgs[:len(smsgs) // STEPS]]
    assert kinds.count('reduce_to_owner') == world and kinds.count('gather_blocks') == 2
    assert sum(m[1] for m in smsgs[:len(smsgs) // STEPS] if m[0] == 'reduce_to_owner') == 8192 * 25088
    # ---- one process playing the ranks in turn
    eng, seq_losses = _sequential_reference(dev, world)
    got_losses = np.concatenate([np.load(str(tmp_path / ('losses%d.npy' % r))) for r in range(world)], 1)
    assert np.array_equal(got_losses[0], seq_losses[0])              # same kernels, same inputs
    assert np.allclose(got_losses, seq_losses, rtol=1e-5, atol=1e-7)
    seq = eng.params.cpu().numpy()
    step = np.abs(seq - synthetic_flat(eng)).max()
    assert step > 0
    d = np.abs(ref['params'] - seq).max()
    ds = np.abs(sref['params'] - seq).max()
    dr = np.abs(sref['params'] - ref['params']).max()
    print('\n[world %d] max |params - sequential reference|: all-reduce route %.3e, sharded route '
          '%.3e; sharded vs all-reduce %.3e (total movement %.3e)' % (world, d, ds, dr, step))
    assert d <= 1e-3 * step + 1e-9, (d, step)
    assert ds <= 1e-3 * step + 1e-9, (ds, step)
    assert dr <= 1e-3 * step + 1e-9, (dr, step)
    # the planes every rank holds are a valid split of the parameters it holds: fc6_w rebuilt
    # from hi + lo planes x 1/scale is within 2^-21 of a row maximum of the fp32 master rows
    n6, k6 = 8192, 25088
    w6 = sref['params'][:n6 * k6].reshape(n6, k6)
    pl = sref['planes'].view(np.float16).astype(np.float32)           # [2, k6/16, n6, 16]
    rows = np.arange(0, n6, n6 // world // 4)                         # every owner's block sampled
    rebuilt = (pl[0][:, rows] + pl[1][:, rows]).transpose(1, 0, 2).reshape(len(rows), k6) \
        * sref['scales'][rows, None]
    err = np.abs(rebuilt - w6[rows]).max(axis=1) / np.abs(w6[rows]).max(axis=1)
    assert err.max() <= 2.0 ** -20, err.max()
    del eng
    torch.cuda.empty_cache()
    # ---- one process, all the images as one batch
    from detectron.datasets import synthetic
    roidb, _shares = _batches(world)
    eng = _engine(dev, world * B)
    one_losses = _run(eng, synthetic.make_minibatch(roidb, C), dev)
    assert np.allclose(got_losses[0], one_losses[0], rtol=1e-6, atol=0)
    assert np.allclose(got_losses, one_losses, rtol=2e-4, atol=1e-6)
    d1 = np.abs(ref['params'] - eng.params.cpu().numpy()).max()
    assert d1 <= 1e-3 * step + 1e-9, (d1, step)


def test_pipelined_update_bit_identical_to_the_unpipelined_route(dev, tmp_path):
    """NAWS.PIPELINE_UPDATE on hardware (two ranks on one GPU over gloo), Dropout ON: fc6's biases
    as the first message, fc6_w updated in two row pieces as its chunks arrive, the next
    iteration's fc6 forward launched piece by piece behind them with the full launch's Dropout
    counters.  After three training steps the losses of every step, the parameters, the momentum,
    fc6_w's operand planes and their scales are bit-identical to the route that waits for the whole
    exchange and updates in one launch, on both ranks."""
    ctx = mp.get_context('spawn')
    # 'u': one update launch after the whole exchange; 'p': piece by piece, queued by sgd_step();
    # 'e': piece by piece, queued from inside backward (train_step, what the training loop calls)
    for pipeline, tag, ts in ((False, 'u', False), (True, 'p', False), (True, 'e', True)):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), False, pipeline, 0.5, tag, ts))
                 for r in range(2)]
        _run_ranks(procs)
    import json
    mu = json.load(open(str(tmp_path / 'messagesu.json')))
    mp_ = json.load(open(str(tmp_path / 'messagesp.json')))
    assert len(mu) // STEPS == 5 and len(mp_) // STEPS == 6          # 4 chunks + small | + the biases
    assert sum(m[1] for m in mu) == sum(m[1] for m in mp_)           # same bytes
    assert json.load(open(str(tmp_path / 'messagese.json'))) == mp_   # same messages, same order
    for what in ('params', 'mom', 'planes', 'scales', 'losses'):
        for r in range(2):
            a = np.load(str(tmp_path / ('%su%d.npy' % (what, r))))
            for t in ('p', 'e'):
                b = np.load(str(tmp_path / ('%s%s%d.npy' % (what, t, r))))
                assert np.array_equal(a, b), (what, t, r)
    assert np.isfinite(np.load(str(tmp_path / 'lossesp0.npy'))).all()
