"""The C-ABI library loads on a CPU-only box and exports every symbol include/naws.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'naws.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(naws_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_table_agree():
    from naws_hip import lib
    assert _declared() == lib.ALL_SYMBOLS


def test_library_exports_every_declared_symbol():
    from naws_hip import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    l = ctypes.CDLL(lib.LIB_PATH)
    for name in _declared():
        assert hasattr(l, name), name
    assert b'gfx950' in lib.load().naws_version()


def test_ops_refuse_cpu_tensors():
    import torch
    from naws_hip import ops
    with pytest.raises(TypeError):
        ops.roi_iou(torch.zeros((4, 5)))
    with pytest.raises(TypeError):
        ops.gemm(torch.zeros((8, 8)), torch.zeros((8, 8)))


def test_product_never_imports_oracle():
    bad = []
    for base, _d, files in os.walk(os.path.join(ROOT, 'na-fwebsod_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
                    bad.append(f)
    assert not bad, bad
