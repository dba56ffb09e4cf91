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


def _batches():
    """Four images' worth of inputs: rank r takes images 2r, 2r + 1 (make_minibatch numbers the
    images of a batch from 0, so each half is built on its own)."""
    from detectron.datasets import synthetic
    roidb = synthetic.make_roidb(2 * B, 48, C, 96, 128, seed=5)
    halves = [synthetic.make_minibatch(roidb[r * B:(r + 1) * B], C) for r in range(2)]
    return roidb, halves


def _engine(dev, gpu_num, pg=None, world=1, sharded=False):
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    eng = WsddnEngine(C + 1, dev, dropout=0.0, gpu_num=gpu_num, seed=5, process_group=pg,
                      world_size=world, allreduce_chunks=4, sharded_update=sharded)
    blobs = synthetic.init_blobs(C, seed=5)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    eng.set_lr(LR)
    return eng


def _run(eng, mb, dev):
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    n = int(mb['data'].shape[0])
    seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=n)).tolist()
    losses = []
    for _ in range(STEPS):
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
        eng.sgd_step()
        losses.append(out['loss_cls'].cpu().numpy().copy())
    eng.flush()
    torch.cuda.synchronize()
    return np.stack(losses)


def _worker(rank, world, port, outdir, sharded=False):
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    import faulthandler
    # a rank stuck in a collective says where, then leaves (the parent's limit is 300 s)
    faulthandler.dump_traceback_later(float(os.environ.get('NAWS_RANK_LIMIT', '240')), exit=True)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _roidb, halves = _batches()
    eng = _engine(dev, world * B, dist.group.WORLD, world, sharded=sharded)
    assert eng.reducer.active and (eng._shard_blocks() is not None) == sharded
    losses = _run(eng, halves[rank], dev)
    tag = 's' if sharded else ''
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
    np.save(os.path.join(outdir, 'params%s%d.npy' % (tag, rank)), eng.params.cpu().numpy())
    np.save(os.path.join(outdir, 'mom%s%d.npy' % (tag, rank)), eng.momentum_buf.cpu().numpy())
    np.save(os.path.join(outdir, 'planes%s%d.npy' % (tag, rank)),
            wp['w6'].planes.view(torch.int16).cpu().numpy())
    np.save(os.path.join(outdir, 'scales%s%d.npy' % (tag, rank)), wp['w6'].inv_scale.cpu().numpy())
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
