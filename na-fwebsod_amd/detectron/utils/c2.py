"""Initializer shorthands used by the model builders (reference: detectron/utils/c2.py)."""


def const_fill(value):
    return ('ConstantFill', {'value': value})


def gauss_fill(std):
    return ('GaussianFill', {'mean': 0.0, 'std': std})
