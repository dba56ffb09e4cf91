"""Training statistics: median-smoothed losses / metrics and `json_stats: {...}` lines in the
reference's format (detectron/utils/training_stats_wsl.py:27-98, logging.py:32-38) so the
upstream log parsers (tools/visualize_learn_*.py) keep working."""
import collections
import datetime
import json
import time

import numpy as np

from detectron.core.config import cfg


class SmoothedValue(object):
    def __init__(self, window):
        self.deque = collections.deque(maxlen=window)
        self.total, self.count = 0.0, 0

    def AddValue(self, v):
        self.deque.append(v)
        self.total += v
        self.count += 1

    def GetMedianValue(self):
        return float(np.median(self.deque))

    def GetGlobalAverageValue(self):
        return self.total / max(self.count, 1)


def log_json_stats(stats, printer=print):
    printer('json_stats: {:s}'.format(json.dumps(stats, sort_keys=True)))


class TrainingStats(object):
    def __init__(self, model, printer=print):
        self.LOG_PERIOD = max(1, int(1280 / cfg.NUM_GPUS))
        self.WIN_SZ = self.LOG_PERIOD
        self.model, self.printer = model, printer
        self.smoothed_losses_and_metrics = {
            k: SmoothedValue(self.WIN_SZ) for k in model.losses + model.metrics}
        self.smoothed_total_loss = SmoothedValue(self.WIN_SZ)
        self.iter_total_loss = np.nan
        self.iter_time = SmoothedValue(self.WIN_SZ)
        self._tic = None

    def IterTic(self):
        self._tic = time.time()

    def IterToc(self):
        self.iter_time.AddValue(time.time() - self._tic)

    def ResetIterTimer(self):
        self.iter_time = SmoothedValue(self.WIN_SZ)

    def UpdateIterStats(self, values):
        """values: {loss or metric name: float} for this iteration (already averaged over
        this process's images and, by the caller, over ranks: net_wsl.py:210-220)."""
        total = 0.0
        for k, v in values.items():
            if k in self.smoothed_losses_and_metrics:
                self.smoothed_losses_and_metrics[k].AddValue(v)
            if k in self.model.losses:
                total += v
        self.iter_total_loss = total
        self.smoothed_total_loss.AddValue(total)

    def LogIterStats(self, cur_iter, lr, queue_size=0, mem_mb=0):
        if cur_iter % self.LOG_PERIOD == 0 or cur_iter == cfg.SOLVER.MAX_ITER - 1:
            log_json_stats(self.GetStats(cur_iter, lr, queue_size, mem_mb), self.printer)

    def GetStats(self, cur_iter, lr, queue_size=0, mem_mb=0):
        eta = self.iter_time.GetGlobalAverageValue() * (cfg.SOLVER.MAX_ITER - cur_iter)
        stats = dict(iter=cur_iter, lr=float(lr), time=self.iter_time.GetGlobalAverageValue(),
                     loss=self.smoothed_total_loss.GetMedianValue(),
                     eta=str(datetime.timedelta(seconds=int(eta))), mb_qsize=int(queue_size),
                     mem=int(mem_mb))
        for k, v in self.smoothed_losses_and_metrics.items():
            stats[k] = v.GetMedianValue()
        return stats
