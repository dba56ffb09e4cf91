#!/usr/bin/env python3
"""Golden fixture for the training log: the reference's own `TrainingStats` + `log_json_stats`
(`detectron/utils/training_stats_wsl.py:22-96`, `detectron/utils/logging.py:32-38, 41-65`) driven
for 400 iterations with a deterministic clock, a seeded series of losses / metrics / queue sizes
and a fixed memory figure; the printed `json_stats: {...}` lines (NUM_GPUS = 8: LOG_PERIOD = WIN_SZ
= 160) are written to reference_training_stats.json with the inputs that produced them.

Mocks: caffe2 (MagicMock; `GetGPUMemoryUsageStats` returns the fixed figure),
`workspace.FetchBlob` (this iteration's float32 scalar of `gpu_i/<blob>`; the reference's own
`net_wsl.average_multi_gpu_blob` averages them), the model (losses
/ metrics names and a loader whose queue reports this iteration's size), `time.time`.

Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_training_stats.py
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_from_reference import REF, _StubFinder  # noqa: E402

LOSSES = ['loss_cls', 'loss_cls_noise']
METRICS = ['accuracy_cls', 'accuracy_cls_noise']
ITERS, MAX_ITER, NUM_GPUS = 400, 400, 8


def series(seed=3):
    """The per-iteration inputs: values of the four blobs, queue size, lr, seconds per iteration."""
    rng = np.random.default_rng(seed)
    decay = np.exp(-np.arange(ITERS) / 300.0)[:, None]
    vals = {k: (rng.uniform(0.05, 1.5, (ITERS, NUM_GPUS)) * decay).astype(np.float32) for k in LOSSES}
    vals.update({k: rng.integers(0, 2, (ITERS, NUM_GPUS)).astype(np.float32) for k in METRICS})
    qsize = rng.integers(0, 65, ITERS)
    lr = np.where(np.arange(ITERS) < 250, np.float32(1e-3), np.float32(1e-4)).astype(np.float32)
    dt = rng.uniform(0.010, 0.020, ITERS)
    return vals, qsize, lr, dt


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', NUM_GPUS, 'SOLVER.MAX_ITER', MAX_ITER])
    import detectron.utils.training_stats_wsl as ts
    import detectron.utils.timer as timer_mod

    vals, qsize, lr, dt = series()
    cur = {'it': 0, 'clock': 1000.0}
    # the reference's own net_wsl.average_multi_gpu_blob runs: only the workspace fetch is replaced
    def fetch(name):
        gpu, blob = name.split('/', 1)
        return np.array(vals[blob][cur['it']][int(gpu[4:])], dtype=np.float32)
    ts.nu.workspace.FetchBlob = fetch
    mem_bytes = 7 * 1024 ** 3 + 12345
    ts.c2_py_utils.GetGPUMemoryUsageStats = lambda: {'max_by_gpu': np.array([mem_bytes] * NUM_GPUS)}
    timer_mod.time.time = lambda: cur['clock']

    class _Q(object):
        def qsize(self):
            return int(qsize[cur['it']])

    class _Loader(object):
        _minibatch_queue = _Q()

    class _Model(object):
        losses, metrics = LOSSES, METRICS
        roi_data_loader = _Loader()

    stats = ts.TrainingStats(_Model())
    lines = []
    for it in range(ITERS):
        cur['it'] = it
        stats.IterTic()
        cur['clock'] += float(dt[it])
        stats.IterToc()
        stats.UpdateIterStats()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            stats.LogIterStats(it, lr[it])
        for l in buf.getvalue().splitlines():
            lines.append([it, l])
        if it == stats.LOG_PERIOD:                 # train_wsl.py:72-75: the timer restarts once
            stats.ResetIterTimer()
    out = dict(NUM_GPUS=NUM_GPUS, MAX_ITER=MAX_ITER, LOG_PERIOD=stats.LOG_PERIOD, WIN_SZ=stats.WIN_SZ,
               losses=LOSSES, metrics=METRICS, mem_bytes=mem_bytes,
               values={k: [[float(x) for x in row] for row in v] for k, v in vals.items()},
               qsize=[int(x) for x in qsize], lr=[float(x) for x in lr], dt=[float(x) for x in dt],
               lines=lines)
    with open(os.path.join(HERE, 'reference_training_stats.json'), 'w') as f:
        json.dump(out, f)
    for it, l in lines:
        print(it, l)


if __name__ == '__main__':
    main()
