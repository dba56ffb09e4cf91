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


def test_row_chunks_cover():
    from naws_hip.reducer import row_chunks
    for rows, n in [(8192, 8), (8192, 3), (100, 8), (128, 1)]:
        ch = row_chunks(rows, n)
        assert ch[0][0] == 0 and ch[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all((b - a) % 128 == 0 for a, b in ch[:-1]) and len(ch) <= n
