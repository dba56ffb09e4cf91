"""Environment helpers the test engine uses (reference: detectron/utils/env.py:30-91)."""
import os
import sys

import yaml


def get_runtime_dir():
    """Directory holding the tool scripts (reference: env.py:36-38 returns its tools/ dir)."""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                        'tools')


def get_py_bin_ext():
    """The tools are plain scripts (reference: env.py:41-43)."""
    return '.py'


def _plain(node):
    """Plain yaml types: ndarray / tuple values become lists, which the reference's type check
    converts back when it merges them (config.py:1393-1420), as does this package's."""
    import numpy as np
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, np.ndarray):
        return node.tolist()
    if isinstance(node, (tuple, list)):
        return [_plain(v) for v in node]
    if isinstance(node, np.generic):
        return node.item()
    return node


REFERENCE_NODE_TAG = 'tag:yaml.org,2002:python/object/new:detectron.utils.collections.AttrDict'


class _RefNode(dict):
    """A mapping that dumps as the reference's AttrDict does."""


class _RefDumper(yaml.Dumper):
    pass


def _represent_ref_node(dumper, data):
    # what yaml emits for a dict subclass with instance state (detectron/utils/collections.py:
    # AttrDict keeps `__immutable__` in its __dict__): {dictitems: ..., state: ...}
    return dumper.represent_mapping(REFERENCE_NODE_TAG, {'dictitems': dict(data),
                                                        'state': {'__immutable__': False}})


_RefDumper.add_representer(_RefNode, _represent_ref_node)


def _as_ref_nodes(tree):
    return _RefNode((k, _as_ref_nodes(v) if isinstance(v, dict) else v) for k, v in tree.items())


def yaml_dump(cfg_node, stream=None, reference_format=False):
    """The cfg tree as yaml.

    Default: a plain mapping (this package's `load_cfg`, the range-config snapshot handed to the
    children of a multi-GPU test).

    reference_format: the text a reference-side reader takes - tools/reval.py:88-93 and
    utils/net_wsl.py:64-66 run `load_cfg` (the unsafe yaml loader) on the `cfg` string of a
    detections / weights pickle and then use the result AS AN AttrDict (`merge_cfg_from_cfg`
    asserts the type, `configure_bbox_reg_weights` reads `saved_cfg.MODEL`), so every mapping
    carries the tag the reference's own dump gives it (env.py:91 yaml.dump of an AttrDict:
    `!!python/object/new:detectron.utils.collections.AttrDict {dictitems, state}`); ndarray /
    tuple values go out as lists, which the reference's type check converts back
    (config.py:1393-1420); the NAWS subtree - options of this implementation, unknown keys to
    the reference's merge - is left out."""
    tree = _plain(cfg_node)
    if not reference_format:
        return yaml.dump(tree, stream=stream)
    tree = {k: v for k, v in tree.items() if k != 'NAWS'}
    return yaml.dump(_as_ref_nodes(tree), stream=stream, Dumper=_RefDumper)


def python_binary():
    return sys.executable
