"""Learning-rate schedule.  Same entry point and policies as the reference
(detectron/utils/lr_policy.py:28-131): `get_lr_at_iter(it)` -> np.float32."""
import numpy as np

from detectron.core.config import cfg


def get_step_index(cur_iter):
    """Index of the lr step `cur_iter` falls in (SOLVER.STEPS must start at 0)."""
    assert cfg.SOLVER.STEPS[0] == 0, 'The first step should always start at 0.'
    bounds = list(cfg.SOLVER.STEPS) + [cfg.SOLVER.MAX_ITER]
    ind = 0
    for ind, b in enumerate(bounds):
        if cur_iter < b:
            break
    return ind - 1


def lr_func_steps_with_lrs(cur_iter):
    return cfg.SOLVER.LRS[get_step_index(cur_iter)]


def lr_func_steps_with_decay(cur_iter):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** get_step_index(cur_iter)


def lr_func_step(cur_iter):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** (cur_iter // cfg.SOLVER.STEP_SIZE)


def lr_func_cosine_decay(cur_iter):
    frac = float(cur_iter) / cfg.SOLVER.MAX_ITER
    return cfg.SOLVER.BASE_LR * 0.5 * (np.cos(np.pi * frac) + 1)


def lr_func_exp_decay(cur_iter):
    frac = float(cur_iter) / cfg.SOLVER.MAX_ITER
    return cfg.SOLVER.BASE_LR * np.exp(frac * np.log(cfg.SOLVER.GAMMA))


_POLICIES = {
    'steps_with_lrs': lr_func_steps_with_lrs, 'steps_with_decay': lr_func_steps_with_decay,
    'step': lr_func_step, 'cosine_decay': lr_func_cosine_decay, 'exp_decay': lr_func_exp_decay,
}


def get_lr_func():
    if cfg.SOLVER.LR_POLICY not in _POLICIES:
        raise NotImplementedError('Unknown LR policy: {}'.format(cfg.SOLVER.LR_POLICY))
    return _POLICIES[cfg.SOLVER.LR_POLICY]


def get_lr_at_iter(it):
    lr = get_lr_func()(it)
    if it < cfg.SOLVER.WARM_UP_ITERS:
        method = cfg.SOLVER.WARM_UP_METHOD
        if method == 'constant':
            factor = cfg.SOLVER.WARM_UP_FACTOR
        elif method == 'linear':
            alpha = it / cfg.SOLVER.WARM_UP_ITERS
            factor = cfg.SOLVER.WARM_UP_FACTOR * (1 - alpha) + alpha
        else:
            raise KeyError('Unknown SOLVER.WARM_UP_METHOD: {}'.format(method))
        lr *= factor
    return np.float32(lr)
