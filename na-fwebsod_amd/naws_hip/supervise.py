"""Per-rank supervisor of a multi-rank bench / training job (no torch, no HIP: this process never
touches the GPU, so it may start, watch, kill and RESTART the process that does).

Why it exists: the piece-by-piece deferred update (NAWS.PIPELINE_UPDATE) and the sharded update
issue collectives in patterns that were validated over gloo and one-rank RCCL only - no multi-GPU
node was ever available to this project.  The first N > 1 RCCL run may be the only one; a hang in
it must cost a bounded number of seconds and end in a number on the one-launch route, not in a
lost lease.  The reference has no counterpart (its ranks are threads of one Caffe2 workspace,
detectron/modeling/optimizer_wsl.py:52-72, detectron/utils/train_wsl.py:52-104).

Shape (one node, launched by torch.distributed.run or by `bench.py --gpus N` itself):

    torchrun ── rank r: SUPERVISOR (this module; RANK / WORLD_SIZE in its environment)
                          └── WORKER (the same command line + NAWS_BENCH_WORKER=1): all GPU work

* The worker appends one JSON line per PHASE to its heartbeat file (`Heartbeat.phase`).
* The supervisor polls: worker exit code, the age of the worker's last heartbeat against that
  phase's deadline, and the attempt's shared failure marker (any rank's supervisor that sees its
  worker die or stall writes it; every supervisor that sees it kills its own worker) - so all
  ranks leave an attempt together, within a poll interval.
* Failure BEFORE the main result exists -> next attempt of the `ladder` (fresh workers, a fresh
  rendezvous through a FileStore in the shared directory; the reason travels to the new workers in
  NAWS_BENCH_FALLBACK and ends up in the result line as `route_fallback`).
* Failure AFTER rank 0 saved the main result (the in-run A/B legs, the extras): the supervisors
  kill the workers, rank 0's supervisor prints the saved line with `ab_failed` set, all exit 0.

Nothing here re-executes a process that has initialised the GPU: workers are children, started
fresh, and the supervisor exits with the last worker's code."""
import json
import os
import signal
import subprocess
import sys
import time

# seconds a worker may stay in a phase without a new heartbeat (fixed part, per-step part):
# generous against false alarms (a fresh box pages torch in for 1-2 minutes; RCCL builds its
# rings on first use), bounded so that a hang costs minutes, not the lease
DEADLINES = {
    'boot': (600.0, 0.0),            # process start -> first heartbeat (import torch)
    'init': (300.0, 0.0),            # process group + first collective
    'setup': (300.0, 0.0),           # engine, blobs, parameter broadcast
    'allreduce_alone': (180.0, 0.0),
    'warmup': (240.0, 2.0),          # x warm-up steps (first steps: plane splits, RCCL channels)
    'digest': (120.0, 0.0),
    'timed': (90.0, 1.0),            # x timed steps
    'main_done': (120.0, 0.0),
    'ab_unpipelined': (120.0, 1.5),
    'ab_pipelined': (120.0, 1.5),
    'ab_sharded': (120.0, 1.5),
    'ab_one_message': (120.0, 1.5),
    'extras': (900.0, 0.0),          # rank 0's conv-alone / RoIPool-alone timings
    'emit': (120.0, 0.0),
    'done': (120.0, 0.0),            # destroy_process_group, interpreter exit
}
POST_MAIN = ('main_done', 'ab_unpipelined', 'ab_pipelined', 'ab_sharded', 'ab_one_message', 'extras',
             'emit', 'done')
EXIT_DIGEST_MISMATCH = 4


class Heartbeat(object):
    """Worker side.  path None (no supervisor) -> every call is a no-op."""

    def __init__(self, path=None):
        self.path = path if path is not None else os.environ.get('NAWS_BENCH_HB')
        self.t0 = time.time()

    def phase(self, name, steps=0, **info):
        if not self.path:
            return
        rec = dict(phase=name, t=time.time(), steps=int(steps))
        rec.update(info)
        with open(self.path, 'a') as f:
            f.write(json.dumps(rec) + '\n')
            f.flush()


def deadline_of(rec, scale=1.0):
    fixed, per = DEADLINES.get(rec.get('phase'), (300.0, 0.0))
    return scale * (fixed + per * rec.get('steps', 0))


def last_heartbeat(path, started):
    """The newest record of a heartbeat file; before the first one: phase 'boot' since `started`."""
    try:
        with open(path) as f:
            lines = [l for l in f.read().splitlines() if l.strip()]
        if lines:
            return json.loads(lines[-1])
    except (OSError, ValueError):
        pass
    return dict(phase='boot', t=started, steps=0)


def default_ladder(argv):
    """The routes tried in turn, as extra command-line flags of the worker: the caller's own
    choice first, then the one-launch update behind the whole exchange (round 4's route), then
    the same with fc6_w's gradient as ONE message (no chunking: the plainest exchange there is)."""
    ladder = [('as launched', [])]
    if '--no-pipeline-update' not in argv and '--sharded-update' not in argv:
        ladder.append(('unpipelined', ['--no-pipeline-update']))
    elif '--sharded-update' in argv:
        ladder.append(('unpipelined', ['--no-pipeline-update', '--drop-sharded-update']))
    ladder.append(('one message', ['--no-pipeline-update', '--drop-sharded-update',
                                   '--allreduce-chunks', '1']))
    return ladder


def _proc_start_time(pid):
    try:
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        return '0'


def shared_dir(environ):
    """One directory per JOB, the same for every rank's supervisor: keyed on the launcher
    process (the ranks' common parent: torch.distributed.run's agent) and its start time, plus
    the rendezvous port.  NAWS_BENCH_SHARED overrides it (launchers whose ranks have no common
    parent)."""
    d = environ.get('NAWS_BENCH_SHARED')
    if not d:
        ppid = os.getppid()
        d = os.path.join(environ.get('TMPDIR', '/tmp'), 'naws_bench_%s_%d_%s'
                         % (environ.get('MASTER_PORT', '0'), ppid, _proc_start_time(ppid)))
    os.makedirs(d, exist_ok=True)
    return d


def _die_with_parent():
    """preexec of the worker: SIGKILL it when this supervisor dies without a chance to clean up
    (the launcher's SIGKILL after its SIGTERM grace) - no orphan is left holding a GPU."""
    try:
        import ctypes
        ctypes.CDLL(None).prctl(1, signal.SIGKILL)       # PR_SET_PDEATHSIG
    except Exception:                                    # noqa: BLE001 - best effort, Linux only
        pass


def _kill(proc, grace=5.0):
    """SIGTERM to the worker's process group, SIGKILL after `grace` seconds."""
    if proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGTERM)
    except OSError:
        pass
    t0 = time.time()
    while proc.poll() is None and time.time() - t0 < grace:
        time.sleep(0.05)
    if proc.poll() is None:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()


def _write_once(path, text):
    """Create `path` with `text` unless it exists (the first reporter's reason stands)."""
    try:
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
    except FileExistsError:
        return False
    with os.fdopen(fd, 'w') as f:
        f.write(text)
    return True


def supervise_rank(script, argv, environ=None, ladder=None, poll=0.2, log=None):
    """Run `python script argv...` as this rank's worker under the watchdog; returns the exit
    code for this (supervisor) process.  See the module docstring."""
    environ = dict(os.environ if environ is None else environ)
    rank = int(environ.get('RANK', '0'))
    scale = float(environ.get('NAWS_BENCH_DEADLINE_SCALE', '1.0'))
    ladder = default_ladder(argv) if ladder is None else ladder
    sdir = shared_dir(environ)
    log = log or (lambda m: print('bench.py supervisor[rank %d]: %s' % (rank, m), file=sys.stderr,
                                  flush=True))
    current = {}

    def on_signal(signum, _frame):           # the launcher aborts the job: take the worker along
        if current.get('proc') is not None:
            _kill(current['proc'])
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)

    reason = None
    last_rc = 1
    for k, (name, extra) in enumerate(ladder):
        hb = os.path.join(sdir, 'hb_%d_%d.jsonl' % (k, rank))
        fail = os.path.join(sdir, 'fail_%d.json' % k)
        mainline = os.path.join(sdir, 'main_%d.json' % k)
        env = dict(environ)
        env.update(NAWS_BENCH_WORKER='1', NAWS_BENCH_HB=hb, NAWS_BENCH_ATTEMPT=str(k),
                   NAWS_BENCH_MAINLINE=mainline, NAWS_BENCH_ROUTE=name)
        if k > 0:
            # a fresh rendezvous: the launcher's TCP store still holds the first attempt's keys
            env['NAWS_BENCH_STORE'] = os.path.join(sdir, 'store_%d' % k)
            env['NAWS_BENCH_FALLBACK'] = reason or 'attempt %d failed' % (k - 1)
        cmd = [sys.executable, script] + list(argv) + list(extra)
        if k > 0:
            log('attempt %d (%s): %s' % (k, name, ' '.join(extra)))
        started = time.time()
        proc = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=_die_with_parent)
        current['proc'] = proc
        verdict = None                       # ('ok' | 'fail' | 'post_main', reason)
        while verdict is None:
            rc = proc.poll()
            rec = last_heartbeat(hb, started)
            post = rec.get('phase') in POST_MAIN and os.path.exists(mainline)
            if rc is not None:
                last_rc = rc
                if rc == 0:
                    verdict = ('ok', None)
                    break
                why = ('rank %d: state digests of the ranks differ after warm-up (route %s)'
                       % (rank, name) if rc == EXIT_DIGEST_MISMATCH else
                       'rank %d: worker exited with code %d in phase %s (route %s)'
                       % (rank, rc, rec.get('phase'), name))
                _write_once(fail, json.dumps(dict(reason=why, post_main=bool(post), rank=rank)))
            elif time.time() - rec.get('t', started) > deadline_of(rec, scale):
                why = ('rank %d: no progress for %.0f s in phase %s (route %s)'
                       % (rank, time.time() - rec.get('t', started), rec.get('phase'), name))
                _write_once(fail, json.dumps(dict(reason=why, post_main=bool(post), rank=rank)))
            if os.path.exists(fail):
                try:
                    info = json.load(open(fail))
                except (OSError, ValueError):
                    time.sleep(poll)         # (being written)
                    continue
                _kill(proc)
                verdict = ('post_main' if info.get('post_main') and os.path.exists(mainline)
                           else 'fail', info.get('reason'))
                break
            time.sleep(poll)
        current['proc'] = None
        if verdict[0] == 'ok':
            return 0
        log('%s' % verdict[1])
        if verdict[0] == 'post_main':
            # the headline exists: report it, say which leg did not finish (unless the worker
            # printed its line itself and only hung on its way out)
            if rank == 0 and not last_heartbeat(hb, started).get('printed'):
                try:
                    res = json.load(open(mainline))
                except (OSError, ValueError):
                    return last_rc or 1
                res['ab_failed'] = verdict[1]
                res.setdefault('config', {})['ab_failed'] = verdict[1]
                sys.stdout.flush()
                print(json.dumps(res), flush=True)
            return 0
        reason = verdict[1]
    log('every route failed; last: %s' % reason)
    return last_rc if last_rc not in (0, None) else 1
