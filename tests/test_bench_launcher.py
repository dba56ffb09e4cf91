"""bench.py started bare with --gpus N must start its own ranks: the launcher's
argv / environment, and the whole spawn -> torch.distributed.run -> rendezvous -> one JSON line
path on CPU through `--dry-run` (gloo, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_is_the_drivers_own_shape():
    cmd = bench.launcher_command(8, 29511, ['--gpus', '8', '--steps', '20', '--self-launch',
                                            '--warmup', '5'], python='python3')
    assert cmd[:4] == ['python3', '-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29511'
    tail = cmd[cmd.index(os.path.join(ROOT, 'bench.py')) + 1:]
    assert tail == ['--gpus', '8', '--steps', '20', '--warmup', '5']    # --self-launch dropped


def test_launcher_env_drops_stale_rank_variables():
    env = bench.launcher_env({'RANK': '3', 'WORLD_SIZE': '4', 'LOCAL_RANK': '3', 'MASTER_PORT': '1',
                              'PATH': '/bin', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    assert 'RANK' not in env and 'WORLD_SIZE' not in env and 'MASTER_PORT' not in env
    assert env['PATH'] == '/bin' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert bench.launcher_env({})['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_result_line_detection():
    assert bench.is_result_line('{"metric": "x", "value": 1}\n')
    assert not bench.is_result_line('NCCL version 2.22.3+hip7.0')
    assert not bench.is_result_line('{"not": "the line"}')
    assert not bench.is_result_line('{broken')


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bare_gpus_2_starts_two_ranks_and_prints_one_last_line():
    r = _run('--gpus', '2', '--dry-run', '--steps', '5', '--warmup', '1')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    res = json.loads(lines[-1])                       # the LAST line is the result
    assert sum(bench.is_result_line(l) for l in lines) == 1
    assert any('noise on stdout' in l for l in lines[:-1])      # other ranks' output passes through
    assert res['n_gpus'] == 2 and res['config']['rccl_world_size'] == 2
    assert res['config']['launched_by_bench'] is True
    # rank 1 sleeps twice as long per step: min / max over ranks are both reported
    assert res['config']['ms_per_step_rank_max'] >= res['config']['ms_per_step_rank_min'] > 0
    assert res['ms_per_step'] == res['config']['ms_per_step_rank_max']


def test_self_launch_with_one_rank_goes_through_the_same_spawn_path():
    r = _run('--gpus', '1', '--self-launch', '--dry-run', '--steps', '3', '--warmup', '1')
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['n_gpus'] == 1 and res['config']['launched_by_bench'] is True
    assert 'torch.distributed.run' in r.stderr       # the launcher says what it started


def test_failing_rank_makes_the_launcher_exit_nonzero():
    # inside a 1-rank job --gpus 2 is an error in the child: the parent must relay the failure
    r = _run('--gpus', '2', '--dry-run', env={'RANK': '0', 'WORLD_SIZE': '1'})
    assert r.returncode != 0
    assert 'inside a 1-rank job' in r.stderr


def test_launcher_relays_a_dead_rank():
    r = _run('--gpus', '2', '--dry-run', '--steps', '2', env={'NAWS_DRY_RUN_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert not any(bench.is_result_line(l) for l in r.stdout.splitlines())


# ---- the N > 1 skeleton (broadcast, rank digests, bare exchange, route A/B)
# and the per-rank supervisor's fallbacks, on CPU over gloo (--dry-run: DryJob's arena, the real
# message plans, the real ArenaReducer) -------------------------------------------------------
FAST = {'NAWS_BENCH_DEADLINE_SCALE': '0.04'}     # boot 24 s, a timed phase (90 + K) x 0.04 ~ 4 s


def _line(r):
    lines = [l for l in r.stdout.splitlines() if bench.is_result_line(l)]
    assert lines, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1])


def test_two_ranks_broadcast_digest_bare_exchange_and_route_ab():
    r = _run('--gpus', '2', '--dry-run', '--steps', '6', '--warmup', '2')
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    # (d) the ranks are seeded differently: only the broadcast from rank 0 makes them equal
    assert res['params_broadcast_from_rank0'] is True and res['ranks_equal_before_broadcast'] is False
    # (a) digests of every rank's state agree after warm-up (and after every leg)
    assert res['rank_digest_equal'] is True
    # (b) the exchange alone: the step's own messages, bytes = the whole gradient arena
    assert res['allreduce_alone_ms'] > 0 and res['allreduce_busbw_GBps'] > 0
    assert res['allreduce_bytes'] == 4 * (2 * 128 * 48 + 2 * 128 + 2 * 128 * 128 + 2 * 128 + 4 * 5 * 128 + 4 * 5)
    # (c) the other routes in the same job; value stays the default (pipelined) route's
    assert res['config']['pipelined_update'] is True
    assert res['value_unpipelined'] > 0 and res['rank_digest_equal_unpipelined'] is True
    assert res['value_sharded'] > 0 and res['rank_digest_equal_sharded'] is True
    assert res['value_one_message'] > 0 and res['chunks_one_message'] == 1
    assert res['rank_digest_equal_one_message'] is True
    assert res['route_fallback'] is None and res['supervised'] is True
    # flat copies inside config too (the driver's parser keeps flat keys)
    for k in ('value_unpipelined', 'value_sharded', 'rank_digest_equal', 'allreduce_busbw_GBps'):
        assert res['config'][k] == res[k]


def test_three_ranks_cannot_shard_and_say_so():
    r = _run('--gpus', '3', '--dry-run', '--steps', '3', '--warmup', '1')
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    assert res['value_sharded'] is None and 'do not divide' in res['sharded_skipped']
    assert res['value_unpipelined'] > 0


def test_a_stall_on_the_pipelined_route_falls_back_to_fresh_unpipelined_workers():
    """(e): rank 1 stops inside the timed steps of the pipelined route; every rank's supervisor
    kills its worker and starts a FRESH one with --no-pipeline-update; the line says so."""
    r = _run('--gpus', '2', '--dry-run', '--steps', '4', '--warmup', '1',
             env=dict(FAST, NAWS_BENCH_INJECT='stall:pipelined:1:1'))
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    assert res['route_attempt'] == 1 and res['route_of_attempt'] == 'unpipelined'
    # (whichever rank's supervisor notices first reports: rank 0 waits in its collective meanwhile)
    assert 'no progress' in res['route_fallback'] and 'phase timed' in res['route_fallback']
    assert res['config']['pipelined_update'] is False
    assert res['rank_digest_equal'] is True
    # (its A/B leg then tries the pipelined route again, stalls again: the headline survives)
    assert 'ab_failed' in res and 'ab_pipelined' in res['ab_failed']
    assert sum(bench.is_result_line(l) for l in r.stdout.splitlines()) == 1


def test_a_dead_worker_on_the_pipelined_route_falls_back():
    r = _run('--gpus', '2', '--dry-run', '--steps', '4', '--warmup', '2', '--no-route-ab',
             env=dict(FAST, NAWS_BENCH_INJECT='die:pipelined:0:1'))
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    assert res['route_attempt'] == 1 and 'exited with code 9' in res['route_fallback']


def test_diverging_ranks_exit_with_the_digest_code_and_fall_back():
    """(a): rank 1's state walks away on the pipelined route -> every rank exits 4 after warm-up
    (nothing is timed on diverged ranks) -> fresh workers on the unpipelined route."""
    r = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '2', '--no-route-ab',
             env=dict(FAST, NAWS_BENCH_INJECT='diverge:pipelined:1'))
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    assert res['route_attempt'] == 1 and 'digests of the ranks differ' in res['route_fallback']
    assert 'DIFFERENT state' in r.stderr
    # without a supervisor: the job itself exits non-zero and prints no line
    r = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '2', '--no-supervisor',
             env=dict(NAWS_BENCH_INJECT='diverge:any:1'))
    assert r.returncode != 0
    assert not any(bench.is_result_line(l) for l in r.stdout.splitlines())


def test_a_stall_in_an_ab_leg_keeps_the_headline():
    r = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1',
             env=dict(FAST, NAWS_BENCH_INJECT='stall:sharded:0'))
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r)
    assert res['route_attempt'] == 0 and res['route_fallback'] is None
    assert 'ab_sharded' in res['ab_failed'] and res['config']['ab_failed'] == res['ab_failed']
    assert res['ms_per_step'] > 0 and 'value_sharded' not in res


def test_every_route_failing_is_a_nonzero_exit():
    r = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1',
             env=dict(FAST, NAWS_BENCH_INJECT='die:any:1'))
    assert r.returncode != 0
    assert not any(bench.is_result_line(l) for l in r.stdout.splitlines())
    assert 'every route failed' in r.stderr


def test_supervisor_ladder_and_deadlines():
    from naws_hip import supervise as sv
    assert [n for n, _f in sv.default_ladder(['--gpus', '8'])] == ['as launched', 'unpipelined', 'one message']
    assert sv.default_ladder(['--no-pipeline-update'])[1][1][-2:] == ['--allreduce-chunks', '1']
    lad = sv.default_ladder(['--sharded-update'])
    assert '--drop-sharded-update' in lad[1][1] and '--drop-sharded-update' in lad[2][1]
    assert sv.deadline_of(dict(phase='timed', steps=20)) == 110.0
    assert sv.deadline_of(dict(phase='timed', steps=20), scale=0.5) == 55.0
    assert sv.deadline_of(dict(phase='???')) == 300.0
    assert sv.last_heartbeat('/nonexistent/hb', 12.5) == dict(phase='boot', t=12.5, steps=0)
