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
