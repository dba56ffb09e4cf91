"""bench.py's N > 1 entry point on the one GPU of the box: two ranks sharing
cuda:0 over gloo run the REAL schedule (chunked fc6_w exchange, piece-by-piece update) under the
per-rank supervisor - the rank-0 broadcast onto differently seeded ranks, the rank-digest check,
the bare exchange, the in-run route A/B, and an injected stall that must end in fresh workers on
the unpipelined route; and one rank over RCCL (--force-dist) executes the collectives of every
route that gloo only emulates (dist.reduce, all_gather_into_tensor, per-message Work.wait() from
the update stream).  Reference: detectron/modeling/optimizer_wsl.py:52-72,
detectron/utils/net_wsl.py:183-207."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SMALL = ['--rois', '256', '--height', '160', '--width', '256', '--steps', '3', '--warmup', '2',
         '--no-cpu-baseline', '--no-alt-plan', '--no-extra-configs', '--no-projection']


def _bench(*argv, env=None, timeout=840):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if bench.is_result_line(l)]
    return r, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_on_one_gpu_broadcast_digests_bare_exchange_and_route_ab(dev):
    r, res = _bench('--gpus', '2', '--share-gpu', *SMALL)
    assert r.returncode == 0 and res is not None, r.stderr[-4000:]
    assert res['n_gpus'] == 2 and res['config']['shared_gpu'] is True
    assert res['supervised'] is True and res['route_fallback'] is None and res['route_attempt'] == 0
    # rank 1 was seeded differently: the broadcast made the ranks equal, W + K updates kept them so
    assert res['ranks_equal_before_broadcast'] is False and res['params_broadcast_from_rank0'] is True
    assert res['rank_digest_equal'] is True
    assert res['allreduce_alone_ms'] > 0 and res['allreduce_bytes'] == 957677888   # the whole arena
    # the default route is the piece-by-piece update; the other two ran in the same job
    assert res['config']['pipelined_update'] is True
    assert res['config']['exchange_schedule_equals_projection'] is True
    assert res['value_unpipelined'] > 0 and res['rank_digest_equal_unpipelined'] is True
    assert res['value_sharded'] > 0 and res['rank_digest_equal_sharded'] is True
    assert res['value_one_message'] > 0 and res['chunks_one_message'] == 1
    assert res['rank_digest_equal_one_message'] is True
    assert 'ab_failed' not in res


def test_a_stalled_rank_on_the_pipelined_route_ends_in_fresh_workers_on_the_unpipelined_route(dev):
    r, res = _bench('--gpus', '2', '--share-gpu', '--no-route-ab', *SMALL,
                    env=dict(NAWS_BENCH_INJECT='stall:pipelined:1', NAWS_BENCH_DEADLINE_SCALE='0.15'))
    assert r.returncode == 0 and res is not None, r.stderr[-4000:]
    assert res['route_attempt'] == 1 and res['route_of_attempt'] == 'unpipelined'
    assert 'no progress' in res['route_fallback'] and 'phase timed' in res['route_fallback']
    assert res['config']['pipelined_update'] is False and res['rank_digest_equal'] is True
    assert res['value'] > 0
    assert 'attempt 1 (unpipelined)' in r.stderr


def test_one_rank_over_rccl_executes_every_routes_collectives(dev):
    """--force-dist: world 1 on the nccl backend.  The all-reduce route's chunked messages with
    per-message waits from the update stream (pipelined), the one-launch route, and the sharded
    route's dist.reduce + all_gather_into_tensor legs all execute on RCCL."""
    r, res = _bench('--gpus', '1', '--force-dist', *SMALL)
    assert r.returncode == 0 and res is not None, r.stderr[-4000:]
    assert res['config']['rccl_backend'] == 'nccl' and res['config']['rccl_world_size'] == 1
    assert res['rank_digest_equal'] is True and res['params_broadcast_from_rank0'] is True
    assert res['config']['pipelined_update'] is True
    assert res['value_unpipelined'] > 0 and res['value_sharded'] > 0
    assert res['rank_digest_equal_sharded'] is True
    assert res['allreduce_alone_ms'] > 0
