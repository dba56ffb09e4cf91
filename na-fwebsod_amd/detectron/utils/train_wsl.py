"""Training loop.  Mirrors detectron/utils/train_wsl.py: `train_model` (:33-102),
`create_model` with auto-resume (:112-189), `setup_model_for_training` (:192-222).
One process per GPU (torchrun): RANK / WORLD_SIZE / LOCAL_RANK from the environment."""
import logging
import os
import re

import numpy as np
import torch

from detectron.core.config import cfg, get_output_dir
from detectron.core.executor import NetExecutor
from detectron.utils import lr_policy
from detectron.utils.training_stats_wsl import TrainingStats
import detectron.modeling.model_builder_wsl as model_builder
import detectron.utils.net_wsl as nu

logger = logging.getLogger(__name__)


def dist_env():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    return rank, world, local


def train_model(roidb=None, max_iter=None, printer=print):
    rank, world, local = dist_env()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group('nccl', device_id=device)
        pg = dist.group.WORLD
    model, weights_file, start_iter, checkpoints, output_dir = create_model()
    if 'final' in checkpoints:
        return checkpoints
    executor = setup_model_for_training(model, weights_file, output_dir, device, pg, world, rank,
                                        roidb)
    stats = TrainingStats(model, printer)
    period = max(1, int(cfg.TRAIN.SNAPSHOT_ITERS / max(cfg.NUM_GPUS, 1)))
    last = cfg.SOLVER.MAX_ITER if max_iter is None else min(cfg.SOLVER.MAX_ITER, max_iter)
    loader = model.roi_data_loader
    # Software pipeline: the batch of iteration i+1 is staged (one pinned copy + the device-side
    # image preparation, on the loader's copy stream) right after iteration i has been enqueued
    # and before its losses are fetched, so the host-side staging hides behind the GPU's work.
    # A loader failure on one rank must stop every rank before the next collective, or the others
    # block in it until the watchdog fires: each rank's "my next batch is staged" flag travels with
    # the per-iteration loss all-reduce (no extra collective, no extra sync), and once before the
    # first iteration.
    batch, ok = stage_batch(loader, device) if start_iter < last else (None, True)
    if not agree_ok(ok, pg, world, device):
        handle_critical_error(model, 'roi_data_loader failed' if not ok else
                              'roi_data_loader failed on another rank')
    # The scalars of iteration i are read while iteration i+1 runs (cfg.NAWS.LAGGED_STATS): fetched
    # right away, the read is a host sync per iteration - the GPU idles until the host has queued
    # the next step, and the deferred parameter update cannot hide under the next conv body.  The
    # values, the log lines and their iteration numbers are unchanged; the NaN / failed-loader stop
    # comes one iteration later.
    lag = bool(cfg.NAWS.LAGGED_STATS)

    def account(it, it_lr, handle, my_ok):
        vals, all_ok = finish_iteration_values(handle)
        if not all_ok:
            handle_critical_error(model, 'roi_data_loader failed' if not my_ok else
                                  'roi_data_loader failed on another rank')
        stats.UpdateIterStats(vals, loader.queue_size())
        if rank == 0:
            mem = torch.cuda.max_memory_allocated(device) if device.type == 'cuda' else 0
            if it == last - 1 and last != cfg.SOLVER.MAX_ITER:      # a shortened run: log its last iteration too
                from detectron.utils.training_stats_wsl import log_json_stats
                log_json_stats(stats.GetStats(it, it_lr, mem), stats.printer)
            else:
                stats.LogIterStats(it, it_lr, mem)
        if np.isnan(stats.iter_total_loss):
            handle_critical_error(model, 'Loss is NaN')

    def run(cur_iter, batch):
        stats.IterTic()
        lr = model.UpdateWorkspaceLr(cur_iter, lr_policy.get_lr_at_iter(cur_iter))
        executor.feed(batch)
        executor.run()
        return lr

    def after(cur_iter):
        stats.IterToc()
        if (cur_iter + 1) % period == 0 and cur_iter > start_iter and executor.engine is not None:
            executor.engine.gather_sharded_state()    # collective under NAWS.SHARDED_UPDATE, else a no-op
            if pg is not None and world > 1:
                # every snapshot: the ranks must hold bit-identical parameters, momentum, operand
                # planes and scales (collective; one small all-reduce of exact digests) - a rank
                # that walked away would otherwise train on silently
                from naws_hip.reducer import ranks_agree
                same, bad = ranks_agree(executor.engine.state_tensors(), pg, rank, world)
                if not same:
                    handle_critical_error(model, 'ranks hold different state at iteration %d: %s '
                                          '(NAWS.PIPELINE_UPDATE False / NAWS.SHARDED_UPDATE False '
                                          'select the one-launch all-reduce route)'
                                          % (cur_iter, ','.join(bad)))
        if (cur_iter + 1) % period == 0 and cur_iter > start_iter and rank == 0:
            checkpoints[cur_iter] = os.path.join(output_dir, 'model_iter{}.pkl'.format(cur_iter))
            nu.save_model_to_weights_file(checkpoints[cur_iter], model, executor)
        if cur_iter == start_iter + stats.LOG_PERIOD:
            stats.ResetIterTimer()

    pipelined_iterations(start_iter, last, period, lag, (batch, ok), run,
                         lambda: stage_batch(loader, device),
                         lambda ok_: begin_iteration_values(executor, model, pg, world, ok=ok_),
                         account, after)
    if executor.engine is not None:
        executor.engine.flush()
        executor.engine.gather_sharded_state()
    if rank == 0:
        checkpoints['final'] = os.path.join(output_dir, 'model_final.pkl')
        nu.save_model_to_weights_file(checkpoints['final'], model, executor)
    loader.shutdown()
    return checkpoints


def pipelined_iterations(start_iter, last, period, lag, first, run, stage, begin, account,
                         after=None):
    """The loop of train_model over callables (so that tests/test_distributed_cpu.py can drive it
    over gloo): run(i, batch) -> lr enqueues iteration i; stage() -> (batch, ok) stages the next
    batch; begin(ok) -> handle queues the iteration's scalars + the ok flag (one all-reduce);
    account(i, lr, handle, ok) reads them and raises on a failed loader / NaN.

    With `lag` the scalars of iteration i are read after iteration i+1 has been enqueued.  A rank
    whose loader fails while staging the batch of iteration i+1 therefore cannot stop at once:
    the other ranks will enqueue iteration i+1's collectives before any of them reads the flag.
    It runs iteration i+1 as well - on its previous batch, whose results nobody will use - so
    that every rank has issued the same collectives when all of them raise in account()."""
    batch, ok = first
    pending, prev = None, None
    for cur_iter in range(start_iter, last):
        if batch is None:
            if pending is None or prev is None:
                raise RuntimeError('no batch for iteration %d' % cur_iter)
            batch = prev
        lr = run(cur_iter, batch)
        prev = batch
        batch, ok = stage() if cur_iter + 1 < last else (None, True)
        handle = begin(ok)
        if pending is not None:
            account(*pending)
            pending = None
        if lag and cur_iter + 1 < last and (cur_iter + 1) % period != 0:
            pending = (cur_iter, lr, handle, ok)
        else:
            account(cur_iter, lr, handle, ok)
        if after is not None:
            after(cur_iter)


def stage_batch(loader, device):
    """(next per-GPU batch, True), or (None, False) when this rank's loader has failed."""
    try:
        if loader.has_stopped():
            return None, False
        return loader.next_device_batch(device, cfg.NAWS.IMS_PER_GPU), True
    except Exception:  # the loader thread's error is logged where it happened (loader_wsl.py)
        logger.exception('staging the next batch failed')
        return None, False


def agree_ok(ok, pg, world, device):
    """True when every rank passed ok=True."""
    if pg is None or world <= 1:
        return bool(ok)
    import torch.distributed as dist
    t = torch.tensor([0.0 if ok else 1.0], device=device)
    dist.all_reduce(t, group=pg)
    return float(t.item()) == 0.0


def begin_iteration_values(executor, model, pg, world, ok=True):
    """Queue what an iteration reports - its scalar losses averaged over this process's images and
    over ranks (the reference averages the per-GPU scalars on the host, net_wsl.py:210-220), the
    "every rank passed ok=True" flag riding in the same all-reduce, and this process's accuracies -
    as ONE small device tensor copied to pinned host memory without blocking.  Returns a handle
    for finish_iteration_values."""
    ws = executor.ws
    # The reference averages the per-GPU float32 scalars of EVERY loss and metric on the host in
    # double (sum_multi_gpu_blob: `val += float(blob)`, then / NUM_GPUS).  Here a "GPU" of the
    # reference is one image: per-image values are summed in float64 on the device, the sums travel
    # in ONE all-reduce with the ok flag, and the division by the image count happens once - the
    # same double the reference's loop produces (sums of a few float32 values are exact in float64
    # in any order).
    per = [ws[k].reshape(-1).double() for k in model.losses]
    dev = per[0].device
    n_img = per[0].numel()
    cols = [p.sum() for p in per]
    cols.append(torch.full((), 0.0 if ok else 1.0, device=dev, dtype=torch.float64))
    names = []
    labels = ws['labels_int32'].reshape(-1).to(torch.int64)
    fused = {'accuracy_cls': 'cls_prob', 'accuracy_cls_noise': 'cls_prob_noise'}
    for k in model.metrics:
        if k in ws:
            # the op-by-op plan ran the graph's own Accuracy op (OICR: accuracy_cls1..3 too): one
            # value per process there (one image per process)
            cols.append(ws[k].reshape(-1).double().sum().to(dev) * (n_img / max(ws[k].numel(), 1)))
            names.append(k)
        elif k in fused and fused[k] in ws:
            # the fused engine keeps cls_prob only: top-1 against labels_int32 on the device
            p = ws[fused[k]].reshape(labels.numel(), -1)
            cols.append((p.argmax(1) == labels.to(p.device)).double().sum().to(dev))
            names.append(k)
    t = torch.stack(cols)
    if pg is not None and world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, group=pg)
    nl = len(model.losses)
    den = torch.full_like(t, float(n_img * max(world, 1)))
    den[nl] = 1.0                                     # the flag: a count of failed ranks
    t = t / den                                       # (a division, as the reference's `/ NUM_GPUS`)
    if dev.type == 'cuda':
        host = _stats_slot(t.numel())
        host.copy_(t, non_blocking=True)
        done = torch.cuda.current_stream(dev).record_event()
    else:
        host, done = t.clone(), None
    return list(model.losses), names, host, done


_STATS_RING = {'slots': None, 'next': 0}


def _stats_slot(n):
    """One of a few reusable pinned host vectors (a fresh pinned allocation per iteration costs a
    hipHostMalloc, which synchronises the device; at most two handles are alive at a time)."""
    r = _STATS_RING
    if r['slots'] is None or r['slots'][0].numel() < n:
        r['slots'] = [torch.empty((max(n, 16),), dtype=torch.float64).pin_memory() for _ in range(4)]
        r['next'] = 0
    slot = r['slots'][r['next']]
    r['next'] = (r['next'] + 1) % len(r['slots'])
    return slot[:n]


def finish_iteration_values(handle):
    """-> ({name: float}, all ranks ok) of a begin_iteration_values handle (waits for its copy)."""
    losses, names, host, done = handle
    if done is not None:
        done.synchronize()
    v = host.tolist()
    nl = len(losses)
    out = {k: float(x) for k, x in zip(losses, v[:nl])}
    all_ok = v[nl] == 0.0
    for k, x in zip(names, v[nl + 1:]):
        out[k] = float(x)
    return out, all_ok


def iteration_values(executor, model, pg, world, ok=True):
    """begin + finish in one call (blocking)."""
    return finish_iteration_values(begin_iteration_values(executor, model, pg, world, ok=ok))


def handle_critical_error(model, msg):
    logger.critical(msg)
    model.roi_data_loader.shutdown()
    raise Exception(msg)


def create_model():
    """Build the training model; with TRAIN.AUTO_RESUME pick up the newest model_iter*.pkl."""
    start_iter, checkpoints = 0, {}
    output_dir = get_output_dir(cfg.TRAIN.DATASETS, training=True)
    weights_file = cfg.TRAIN.WEIGHTS
    if cfg.TRAIN.AUTO_RESUME:
        final_path = os.path.join(output_dir, 'model_final.pkl')
        if os.path.exists(final_path):
            logger.info('model_final.pkl exists; no need to train!')
            return None, None, None, {'final': final_path}, output_dir
        best = -1
        for f in os.listdir(output_dir):
            m = re.match(r'model_iter(\d+)\.pkl$', f)
            if m and int(m.group(1)) > best:
                best = int(m.group(1))
                weights_file = os.path.join(output_dir, f)
        if best >= 0:
            start_iter = best + 1
            logger.info('Resuming from checkpoint {} at start iter {}'.format(weights_file, start_iter))
    model = model_builder.create(cfg.MODEL.TYPE, train=True)
    return model, weights_file, start_iter, checkpoints, output_dir


def training_roidb():
    """cfg.TRAIN.DATASETS + cfg.TRAIN.PROPOSAL_FILES -> roidb (roidb_wsl.py:21-58), when the
    named datasets are on disk; otherwise the synthetic roidb of the benchmark (there is no
    dataset on the MI355X image), loudly."""
    from detectron.datasets import dataset_catalog
    names = tuple(cfg.TRAIN.DATASETS)
    if names and all(dataset_catalog.contains(n) and os.path.exists(dataset_catalog.get_ann_fn(n))
                     for n in names):
        from detectron.datasets.roidb_wsl import combined_roidb_for_training
        return combined_roidb_for_training(names, tuple(cfg.TRAIN.PROPOSAL_FILES))
    from detectron.datasets import synthetic
    logger.warning('TRAIN.DATASETS {} not found on disk: using the synthetic roidb'.format(names))
    return synthetic.make_roidb(64, cfg.TRAIN.BATCH_SIZE_PER_IM, cfg.MODEL.NUM_CLASSES - 1,
                                seed=cfg.RNG_SEED)


def setup_model_for_training(model, weights_file, output_dir, device, pg, world, rank, roidb=None):
    executor = NetExecutor(model, device, process_group=pg, world_size=world, rank=rank,
                           images_per_process=cfg.NAWS.IMS_PER_GPU)
    executor.init_params()
    if weights_file and os.path.exists(weights_file):
        nu.initialize_from_weights_file(model, weights_file, executor, broadcast=True)
    else:
        if weights_file:
            logger.warning('weights file {} not found: training from random init'.format(weights_file))
        executor.broadcast_parameters()
    if roidb is None:
        roidb = training_roidb()
    model_builder.add_training_inputs(model, roidb=roidb, rank=rank, world_size=world)
    model.roi_data_loader.start(prefill=False)
    return executor
