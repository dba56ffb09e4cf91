"""Training statistics: window-averaged losses / metrics and `json_stats: {...}` lines exactly as
the reference prints them (detectron/utils/training_stats_wsl.py:22-96, logging.py:32-38, 41-65,
timer.py:35-60), so the upstream log parsers (tools/visualize_learn_*.py) keep working.  Pinned
by tests/golden/reference_training_stats.json: the reference's own classes driven for 400
iterations on a seeded series with a deterministic clock; this module reproduces its lines
character for character (tests/test_config_and_host.py).

What the capture fixed (round 4): the fork logs the WINDOW AVERAGE of every loss / metric
(`GetAverageValue`, not upstream Detectron's median), rounds a window-averaged queue size, takes
`time` / `eta` from a global-average timer, and writes top-level floats as '%.6f' STRINGS.
"""
import collections
import datetime
import json
import time

import numpy as np

from detectron.core.config import cfg


class SmoothedValue(object):
    """logging.py:41-65."""

    def __init__(self, window_size):
        self.deque = collections.deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def AddValue(self, value):
        self.deque.append(value)
        self.count += 1
        self.total += value

    def GetMedianValue(self):
        return np.median(self.deque)

    def GetAverageValue(self):
        return np.mean(self.deque)

    def GetGlobalAverageValue(self):
        return self.total / self.count


class Timer(object):
    """timer.py:35-60."""

    def __init__(self):
        self.reset()

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff

    def reset(self):
        self.total_time, self.calls, self.start_time, self.diff, self.average_time = 0., 0, 0., 0., 0.


def log_json_stats(stats, printer=print, sort_keys=True):
    """logging.py:32-38: top-level floats leave as '%.6f' strings."""
    stats = {k: '{:.6f}'.format(v) if isinstance(v, float) else v for k, v in stats.items()}
    printer('json_stats: {:s}'.format(json.dumps(stats, sort_keys=sort_keys)))


class TrainingStats(object):
    def __init__(self, model, printer=print):
        self.WIN_SZ = max(1, int(1280 / cfg.NUM_GPUS))
        self.LOG_PERIOD = max(1, int(1280 / cfg.NUM_GPUS))
        self.model, self.printer = model, printer
        self.smoothed_losses_and_metrics = {
            k: SmoothedValue(self.WIN_SZ) for k in model.losses + model.metrics}
        self.smoothed_total_loss = SmoothedValue(self.WIN_SZ)
        self.smoothed_mb_qsize = SmoothedValue(self.WIN_SZ)
        self.iter_total_loss = np.nan
        self.iter_timer = Timer()

    def IterTic(self):
        self.iter_timer.tic()

    def IterToc(self):
        return self.iter_timer.toc(average=False)

    def ResetIterTimer(self):
        self.iter_timer.reset()

    def UpdateIterStats(self, values, queue_size=0):
        """values: {loss or metric name: float} for this iteration, already averaged over every
        image of every rank as net_wsl.average_multi_gpu_blob does (train_wsl.begin_iteration_values);
        queue_size: the loader's minibatch queue right now (training_stats_wsl.py:56-69)."""
        for k, v in self.smoothed_losses_and_metrics.items():
            if k in values:
                v.AddValue(values[k])
        self.iter_total_loss = np.sum(np.array([values[k] for k in self.model.losses]))
        self.smoothed_total_loss.AddValue(self.iter_total_loss)
        self.smoothed_mb_qsize.AddValue(queue_size)

    def LogIterStats(self, cur_iter, lr, mem_bytes=0):
        if cur_iter % self.LOG_PERIOD == 0 or cur_iter == cfg.SOLVER.MAX_ITER - 1:
            log_json_stats(self.GetStats(cur_iter, lr, mem_bytes), self.printer)

    def GetStats(self, cur_iter, lr, mem_bytes=0):
        """training_stats_wsl.py:81-96; mem_bytes = the peak device memory (the reference asks
        Caffe2's allocator, the caller here torch's)."""
        eta_seconds = self.iter_timer.average_time * (cfg.SOLVER.MAX_ITER - cur_iter)
        stats = dict(iter=cur_iter, lr=float(lr), time=self.iter_timer.average_time,
                     loss=self.smoothed_total_loss.GetAverageValue(),
                     eta=str(datetime.timedelta(seconds=int(eta_seconds))),
                     mb_qsize=int(np.round(self.smoothed_mb_qsize.GetAverageValue())),
                     mem=int(np.ceil(mem_bytes / 1024 / 1024)))
        for k, v in self.smoothed_losses_and_metrics.items():
            if v.count:
                stats[k] = v.GetAverageValue()
        return stats
