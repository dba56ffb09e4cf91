"""Data-parallel graph construction.  Mirrors detectron/modeling/optimizer_wsl.py:18-137.

The reference replicates the forward graph under `gpu_i/` scopes inside ONE process, adds
one NCCLAllreduce per parameter and one ACMWeightDecayMomentumSGDUpdate per parameter per
GPU.  Here one process drives one MI355X: the forward graph is built once, the gradient
all-reduce is RCCL through torch.distributed (bucketed by the executor), and the update ops
are recorded with the same per-parameter hyper-parameters (biases: lr x2, no weight decay;
'_lrm10_' in the name: lr x10)."""
import logging

from detectron.core.config import cfg
from detectron.modeling.detector import Op

logger = logging.getLogger(__name__)


def build_data_parallel_model(model, single_gpu_build_func):
    if model.only_build_forward_pass or not model.train:
        single_gpu_build_func(model)
        return
    all_loss_gradients = _build_forward_graph(model, single_gpu_build_func)
    model.AddGradientOperators(all_loss_gradients)
    if cfg.NUM_GPUS > 1:
        _add_allreduce_graph(model)
    add_single_gpu_param_update_ops(model, 0)


def _build_forward_graph(model, single_gpu_build_func):
    return dict(single_gpu_build_func(model))


def _add_allreduce_graph(model):
    """One sum-all-reduce per distinct parameter gradient (NCCLAllreduce upstream, :52-72)."""
    model.allreduce_ops = [Op('Allreduce', [model.param_to_grad[p]], [model.param_to_grad[p]],
                              {'backend': 'rccl'}) for p in model.TrainableParams()]


def add_single_gpu_param_update_ops(model, gpu_id):
    lr = model.param_init_net.ConstantFill([], 'lr', shape=[1], value=0.0)
    model.update_ops = []
    for param in model.TrainableParams(gpu_id=gpu_id):
        logger.debug('param ' + str(param) + ' will be updated')
        grad = model.param_to_grad[param]
        acm = model.param_init_net.ConstantFill([param], param + '_acmgrad', value=0.0)
        mom = model.param_init_net.ConstantFill([param], param + '_momentum', value=0.0)
        if param in model.biases:
            weight_decay, lr_mult = 0.0, 2.0
        elif param in model.gn_params:
            weight_decay, lr_mult = 0.0, 1.0
        else:
            weight_decay, lr_mult = cfg.SOLVER.WEIGHT_DECAY, 1.0
        if '_lrm10_' in str(param):
            lr_mult *= 10.0
        model.update_ops.append(Op(
            'ACMWeightDecayMomentumSGDUpdate', [grad, mom, lr, param, acm],
            [grad, mom, param, acm],
            dict(momentum=cfg.SOLVER.MOMENTUM, iter_size=cfg.WSL.ITER_SIZE, gpu_num=cfg.NUM_GPUS,
                 lr_mult=lr_mult, weight_decay=weight_decay)))
