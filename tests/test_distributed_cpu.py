"""The N>1 path on CPU: world_size-2 gloo processes run the gradient all-reduce schedule of
naws_hip.reducer over a CPU arena and agree on the data sharding."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naws_hip.engine import ParamArena, head_param_specs
    from naws_hip.reducer import ArenaReducer, row_chunks
    from detectron.roi_data.loader_wsl import epoch_permutation, rank_shard
    # a small arena with the real blob order (k6 shrunk): rank r holds (r+1) * ramp
    specs = [(n, (s[0] // 32, s[1] // 64) if len(s) == 2 and s[0] == 4096 else s)
             for n, s in head_param_specs(20)]
    arena = ParamArena(specs, torch.device('cpu'))
    ramp = torch.arange(arena.total, dtype=torch.float32) % 97
    g = ramp * (rank + 1)
    red = ArenaReducer(dist.group.WORLD, world)
    red.reduce_async(arena.span(g, 'fc6_b', 'noisy_fc8d_b'))
    w6 = arena.span(g, 'fc6_w', '_[noisy]_fc6_w')
    rows = 2 * specs[0][1][0]
    cols = specs[0][1][1]
    covered = 0
    for r0, r1 in row_chunks(rows, 4, align=32):
        red.reduce_async(w6[r0 * cols:r1 * cols])
        covered += r1 - r0
    red.wait()
    ok = bool(torch.equal(g, ramp * sum(range(1, world + 1)))) and covered == rows
    perm = epoch_permutation(10, 11, 0)
    mine = [int(i) for grp in rank_shard(perm, rank, world) for i in grp]
    q.put((rank, ok, mine, perm.tolist()))
    dist.destroy_process_group()


def test_allreduce_schedule_and_sharding_gloo_world2():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][3] == res[1][3]                                    # same permutation
    assert sorted(res[0][2] + res[1][2]) == sorted(res[0][3])       # disjoint, complete


def _replay_worker(rank, world, port, q):
    """One rank of the engine's iteration on a CPU arena: per-rank gradients, the engine's message
    plan (naws_hip.reducer.message_plan: fc6_w row chunks in the auto chunk count of this world
    size, then the small-gradient message) through ArenaReducer over gloo, the wait the deferred
    update does, then the ACM SGD restatement with gpu_num = world * B."""
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naws_hip.reducer import ArenaReducer, message_plan, message_slice
    arena, params, grads_of, k6, rows6, sgd = _replay_setup()
    B = 2
    g = grads_of(rank)
    red = ArenaReducer(dist.group.WORLD, world)
    chunks = 4 if world == 2 else 2           # engine.py: auto ALLREDUCE_CHUNKS
    plan = message_plan(arena, rows6, chunks, red.active)
    sizes = []
    for kind, rows in plan:
        sl = message_slice(arena, g, kind, rows, k6)
        sizes.append(sl.numel())
        red.reduce_async(sl)
    red.wait()
    p, m = sgd(params.clone(), g, gpu_num=world * B)
    q.put((rank, p.numpy(), m.numpy(), [k for k, _ in plan], sizes))
    dist.destroy_process_group()


def _replay_setup():
    from naws_hip.engine import ParamArena, head_param_specs
    from oracle import oracle
    specs = [(n, (s[0] // 16, s[1] // 64) if len(s) == 2 and s[0] == 4096 else
              ((s[0] // 16,) if s == (4096,) else s)) for n, s in head_param_specs(20)]
    arena = ParamArena(specs, torch.device('cpu'))
    k6 = specs[0][1][1]
    rows6 = 2 * specs[0][1][0]
    gen = torch.Generator().manual_seed(3)
    params = torch.randn((arena.total,), generator=gen)

    def grads_of(rank):
        return torch.randn((arena.total,), generator=torch.Generator().manual_seed(100 + rank))

    def sgd(p, g, gpu_num):
        """optimizer_wsl.py:96-137 per blob (bias: lr x2, no decay) through the oracle's
        restatement of ACMWeightDecayMomentumSGDUpdate; two steps so momentum is exercised."""
        pn, gn = p.numpy().copy(), g.numpy()
        mom = np.zeros_like(pn)
        for name, _shape in specs:
            off, n, _s = arena.offsets[name]
            bias = name.endswith('_b')
            it = 0
            for _ in range(2):
                acm = np.zeros((n,), np.float32)
                pv, mv = pn[off:off + n].copy(), mom[off:off + n].copy()
                it = oracle.acm_sgd(np.ascontiguousarray(gn[off:off + n]), mv,
                                    np.array([1e-2], np.float32), pv, acm, 0.9, 0,
                                    0.0 if bias else 5e-4, 1, gpu_num, 2.0 if bias else 1.0, it)
                pn[off:off + n], mom[off:off + n] = pv, mv
        return torch.from_numpy(pn), torch.from_numpy(mom)
    return arena, params, grads_of, k6, rows6, sgd


def test_engine_message_order_replayed_over_gloo_equals_single_rank():
    """the engine's exact message order (fc6_w row chunks, then the
    small-gradient message), reduced over a world-2 gloo group on a CPU arena and followed by the
    SGD restatement, gives bit-identical parameters and momentum to ONE process that holds both
    ranks' gradients summed and updates with gpu_num = 2 * B."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_replay_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    arena, params, grads_of, k6, rows6, sgd = _replay_setup()
    want_p, want_m = sgd(params.clone(), grads_of(0) + grads_of(1), gpu_num=4)
    for rank, p, m, kinds, sizes in res:
        assert np.array_equal(p, want_p.numpy()) and np.array_equal(m, want_m.numpy()), rank
        # 4 row chunks of fc6_w's gradient first, everything else as ONE message last
        assert kinds == ['fc6_w'] * 4 + ['small']
        assert sum(sizes) == arena.total and sum(sizes[:4]) == rows6 * k6


def _sharded_worker(rank, world, port, q):
    """NAWS.SHARDED_UPDATE on a CPU arena: the engine's message plan with every fc6_w chunk cut
    at the owners' block boundaries and reduced TO ITS OWNER (reduce-scatter, emulated by
    per-owner dist.reduce), the small gradients all-reduced, the SGD restatement on the owner's
    rows only, the updated rows all-gathered (per-owner broadcast on gloo)."""
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naws_hip.reducer import (ArenaReducer, message_plan, message_slice, owner_blocks,
                                  owner_pieces)
    arena, params, grads_of, k6, rows6, sgd = _replay_setup()
    B = 2
    g = grads_of(rank)
    red = ArenaReducer(dist.group.WORLD, world)
    blocks = owner_blocks(rows6, world)
    o6 = arena.offsets['fc6_w'][0]
    gw6 = g[o6:o6 + rows6 * k6].view(rows6, k6)
    plan = message_plan(arena, rows6, 4, red.active)
    legs = []
    for kind, rows in plan:
        if kind == 'fc6_w':
            for o, p0, p1 in owner_pieces(rows[0], rows[1], blocks):
                red.reduce_to_owner_async(gw6[p0:p1].reshape(-1), o)
                legs.append((o, p0, p1))
        else:
            red.reduce_async(message_slice(arena, g, kind, rows, k6))
    red.wait()
    # the update: this rank's fc6_w rows + everything that is not fc6_w
    b0, b1 = blocks[rank]
    p = params.clone()
    pn, mn = sgd(p, g, gpu_num=world * B)           # (non-owned fc6_w rows: garbage in, ignored)
    mine = torch.zeros_like(p, dtype=torch.bool)
    mine[o6 + b0 * k6:o6 + b1 * k6] = True
    mine[o6 + rows6 * k6:] = True
    mine[:o6] = True
    newp = torch.where(mine, pn, p)
    w6 = newp[o6:o6 + rows6 * k6]
    red.gather_blocks_async(w6, rank)
    red.wait()
    mom6 = mn[o6:o6 + rows6 * k6].clone()
    own_mom = mom6[b0 * k6:b1 * k6].clone()
    red.gather_blocks_async(mom6, rank)               # what gather_sharded_state does
    red.wait()
    q.put((rank, newp.numpy(), mom6.numpy(), own_mom.numpy(), legs, (b0, b1)))
    dist.destroy_process_group()


def test_sharded_update_over_gloo_equals_the_allreduce_route():
    """reduce-to-owner + owner-only update + all-gather gives parameters
    (and, once gathered, momentum) bit-identical to the all-reduce route = one process holding
    both ranks' gradients; every fc6_w row travels to exactly one owner."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    arena, params, grads_of, k6, rows6, sgd = _replay_setup()
    want_p, want_m = sgd(params.clone(), grads_of(0) + grads_of(1), gpu_num=4)
    o6 = arena.offsets['fc6_w'][0]
    want_m6 = want_m.numpy()[o6:o6 + rows6 * k6]
    for rank, p, mom6, own_mom, legs, (b0, b1) in res:
        assert np.array_equal(p, want_p.numpy()), rank
        assert np.array_equal(own_mom, want_m6[b0 * k6:b1 * k6]), rank
        assert np.array_equal(mom6, want_m6), rank
        # 4 chunks of 128 rows, 2 owners of 256 rows: every chunk lies inside one owner's block
        assert [l[0] for l in legs] == [0, 0, 1, 1]
        assert sum(l[2] - l[1] for l in legs) == rows6
    assert res[0][5] == (0, rows6 // 2) and res[1][5] == (rows6 // 2, rows6)


def test_owner_blocks_and_pieces():
    from naws_hip.reducer import owner_blocks, owner_pieces
    assert owner_blocks(8192, 8) == [(i * 1024, (i + 1) * 1024) for i in range(8)]
    assert owner_blocks(8192, 3) is None and owner_blocks(96, 2, align=32) is None
    blocks = owner_blocks(8192, 2)
    assert owner_pieces(0, 2048, blocks) == [(0, 0, 2048)]
    assert owner_pieces(2048, 6144, blocks) == [(0, 2048, 4096), (1, 4096, 6144)]
    cover = [pc for r0 in range(0, 8192, 2048) for pc in owner_pieces(r0, r0 + 2048, owner_blocks(8192, 8))]
    assert sum(p1 - p0 for _o, p0, p1 in cover) == 8192 and [c[0] for c in cover] == list(range(8))


def test_row_chunks_cover():
    from naws_hip.reducer import row_chunks
    for rows, n in [(8192, 8), (8192, 3), (100, 8), (128, 1)]:
        ch = row_chunks(rows, n)
        assert ch[0][0] == 0 and ch[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all((b - a) % 128 == 0 for a, b in ch[:-1]) and len(ch) <= n


def _health_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    import types
    from detectron.utils import train_wsl
    pg = dist.group.WORLD
    cpu = torch.device('cpu')
    res = {}
    # before the first iteration: one rank without a batch stops both
    res['pre_all_good'] = train_wsl.agree_ok(True, pg, world, cpu)
    res['pre_rank1_bad'] = train_wsl.agree_ok(rank != 1, pg, world, cpu)
    # inside an iteration: the flag rides in the loss all-reduce, the losses still average
    ex = types.SimpleNamespace(ws={'loss_cls': torch.tensor([1.0 + rank, 3.0 + rank]),
                                   'labels_int32': torch.zeros((1,), dtype=torch.int32)})
    model = types.SimpleNamespace(losses=['loss_cls'], metrics=[])
    vals, ok = train_wsl.iteration_values(ex, model, pg, world, ok=True)
    res['it_good'] = (vals['loss_cls'], ok)
    h = train_wsl.begin_iteration_values(ex, model, pg, world, ok=(rank != 0))     # the lagged form
    vals, ok = train_wsl.finish_iteration_values(h)
    res['it_rank0_bad'] = (vals['loss_cls'], ok)

    class DeadLoader:
        def has_stopped(self):
            return False

        def next_device_batch(self, device, n):
            raise RuntimeError('roi_data_loader failed')
    res['stage'] = train_wsl.stage_batch(DeadLoader(), cpu)
    q.put((rank, res))
    dist.destroy_process_group()


def test_loader_failure_on_one_rank_reaches_every_rank_gloo_world2():
    """a loader failure on one rank used to raise there only and leave the others in
    the next collective.  Both ranks must see it at the same point (detectron/utils/train_wsl.py)."""
    world = 2
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_health_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        r = out[rank]
        assert r['pre_all_good'] is True and r['pre_rank1_bad'] is False
        assert r['it_good'] == (2.5, True)              # mean over both ranks' images
        assert r['it_rank0_bad'] == (2.5, False)
        assert r['stage'] == (None, False)


def _lag_worker(rank, world, port, q, lag):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    import types
    from detectron.utils import train_wsl
    pg = dist.group.WORLD
    log = dict(ran=[], collectives=0, accounted=[], error=None)
    FAIL_AT = 3            # rank 1's loader dies while staging the batch of iteration 3
    staged = [0]

    def stage():
        staged[0] += 1
        if rank == 1 and staged[0] >= FAIL_AT:
            return None, False
        return {'id': staged[0]}, True

    def run(it, batch):
        assert batch is not None                       # feed(None) was the r2 bug
        log['ran'].append((it, batch['id']))
        g = torch.ones(4)
        dist.all_reduce(g, group=pg)                   # the iteration's gradient exchange
        log['collectives'] += 1
        return 1e-3

    def begin(ok):
        ex = types.SimpleNamespace(ws={'loss_cls': torch.tensor([1.0]),
                                       'labels_int32': torch.zeros((1,), dtype=torch.int32)})
        model = types.SimpleNamespace(losses=['loss_cls'], metrics=[])
        log['collectives'] += 1
        return train_wsl.begin_iteration_values(ex, model, pg, world, ok=ok)

    def account(it, lr, handle, my_ok):
        vals, all_ok = train_wsl.finish_iteration_values(handle)
        if not all_ok:
            raise RuntimeError('roi_data_loader failed' if not my_ok else
                               'roi_data_loader failed on another rank')
        log['accounted'].append(it)

    try:
        train_wsl.pipelined_iterations(0, 10, 1000, lag, ({'id': 0}, True), run, stage, begin,
                                       account)
    except RuntimeError as e:
        log['error'] = str(e)
    dist.barrier()                   # no collective is left hanging: both ranks get here
    q.put((rank, log))
    dist.destroy_process_group()


def _run_lag(lag):
    world = 2
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_lag_worker, args=(r, world, port, q, lag)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_lagged_stats_loader_failure_stops_both_ranks_with_matched_collectives():
    """with NAWS.LAGGED_STATS the failing rank used to die in executor.feed(None) while
    the other had already enqueued the next iteration's all-reduce.  Now both ranks issue the
    same collectives and raise at the same iteration."""
    out = _run_lag(True)
    a, b = out[0], out[1]
    assert a['error'] == 'roi_data_loader failed on another rank'
    assert b['error'] == 'roi_data_loader failed'
    assert a['collectives'] == b['collectives']
    assert [i for i, _ in a['ran']] == [i for i, _ in b['ran']] == [0, 1, 2, 3]
    # rank 1 ran iteration 3 on its previous batch (its results are never accounted)
    assert b['ran'][-1][1] == b['ran'][-2][1]
    assert a['accounted'] == b['accounted'] == [0, 1]


def test_unlagged_loader_failure_stops_both_ranks_in_the_same_iteration():
    out = _run_lag(False)
    a, b = out[0], out[1]
    assert a['error'] == 'roi_data_loader failed on another rank'
    assert b['error'] == 'roi_data_loader failed'
    assert a['collectives'] == b['collectives']
    assert [i for i, _ in a['ran']] == [i for i, _ in b['ran']] == [0, 1, 2]


def _pipelined_replay_worker(rank, world, port, q):
    """As _replay_worker with the PIPELINED message order (round 5: fc6's biases first, fc6_w row
    chunks, the rest) and the update's waits in the order engine._apply_update_pipelined issues
    them: wait_first(1) for the biases, wait_first(chunks below each forward piece), wait() for
    the rest - every wait_first must leave exactly the later messages in flight."""
    sys.path.insert(0, os.path.join(ROOT, 'na-fwebsod_amd'))
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naws_hip.reducer import ArenaReducer, message_plan, message_slice
    arena, params, grads_of, k6, rows6, sgd = _replay_setup()
    B = 2
    g = grads_of(rank)
    red = ArenaReducer(dist.group.WORLD, world)
    red.log = log = []
    plan = message_plan(arena, rows6, 4, red.active, pipelined=True)
    covered = torch.zeros((arena.total,), dtype=torch.int32)
    base = g.data_ptr()
    for kind, rows in plan:
        sl = message_slice(arena, g, kind, rows, k6)
        o = (sl.data_ptr() - base) // 4
        covered[o:o + sl.numel()] += 1
        red.reduce_async(sl)
    in_flight = [red.in_flight()]
    red.wait_first(1)                               # the biases
    in_flight.append(red.in_flight())
    chunks = [rows for kind, rows in plan if kind == 'fc6_w']
    arrived = 0
    for piece_end in (rows6 // 2, rows6):           # the two forward pieces
        need = sum(1 for r0, _r1 in chunks if r0 < piece_end)
        red.wait_first(need - arrived)
        arrived = need
        in_flight.append(red.in_flight())
    red.wait()
    in_flight.append(red.in_flight())
    p, m = sgd(params.clone(), g, gpu_num=world * B)
    q.put((rank, p.numpy(), m.numpy(), [k for k, _ in plan], in_flight,
           bool((covered == 1).all()), [n for _k, n in log]))
    dist.destroy_process_group()


def test_pipelined_message_order_over_gloo_equals_the_plain_order():
    """The pipelined plan covers the gradient arena exactly once, starts with fc6's two bias
    vectors, and - summed over gloo at world 2 and updated - gives bit for bit the parameters and
    momentum of the plain order (a + b = b + a: cutting the small message in two changes nothing)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    res = {}
    for target, tag in ((_replay_worker, 'plain'), (_pipelined_replay_worker, 'pipe')):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        out = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[tag] = {o[0]: o for o in out}
    for r in range(2):
        plain, pipe = res['plain'][r], res['pipe'][r]
        assert pipe[3] == ['fc6_b', 'fc6_w', 'fc6_w', 'fc6_w', 'fc6_w', 'rest']
        assert pipe[4] == [6, 5, 3, 1, 0]            # in flight after each wait of the update
        assert pipe[5]                                # every gradient element in exactly one message
        assert pipe[6][0] == 2 * (4096 // 16) and sum(pipe[6]) == sum(plain[4])
        assert np.array_equal(pipe[1], plain[1]) and np.array_equal(pipe[2], plain[2])
    assert np.array_equal(res['pipe'][0][1], res['pipe'][1][1])


def _digest_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naws_hip.reducer import ranks_agree, state_digest
    g = torch.Generator().manual_seed(3)
    state = dict(params=torch.randn((70000,), generator=g), momentum=torch.randn((513,), generator=g),
                 planes=torch.randint(-30000, 30000, (4, 33, 16), generator=g, dtype=torch.int16),
                 words=torch.randint(0, 1 << 30, (129,), generator=g, dtype=torch.int32))
    same0, bad0 = ranks_agree(state, dist.group.WORLD, rank, world)
    # one bit of one float on the last rank; two swapped elements elsewhere (same plain sum)
    if rank == world - 1:
        state['momentum'].view(torch.int32)[7] ^= 1
        p = state['planes'].view(-1)
        p[5], p[900] = p[900].clone(), p[5].clone()
    same1, bad1 = ranks_agree(state, dist.group.WORLD, rank, world)
    d = state_digest([state['planes']])
    q.put((rank, same0, bad0, same1, bad1, d.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_digests_detect_one_flipped_bit_and_a_permutation():
    """naws_hip.reducer.ranks_agree (bench.py's check after warm-up, the training loop's at every
    snapshot): equal buffers agree; one flipped mantissa bit on one rank and a swap of two
    elements (which leaves the plain sum unchanged) are both reported, by buffer name, on EVERY rank."""
    world = 3
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_digest_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same0, bad0, same1, bad1, _d in got:
        assert same0 is True and bad0 == []
        assert same1 is False and bad1 == ['momentum', 'planes']
    assert got[0][5] == got[1][5] and got[0][5][0] == got[2][5][0] and got[0][5][1] != got[2][5][1]
