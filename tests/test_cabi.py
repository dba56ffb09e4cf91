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


def test_argument_validation_happens_before_any_launch():
    """The ENFORCE-style checks (shape / arg / null / unsupported codes, include/naws.h) return
    before a kernel is launched, so they can be exercised without a GPU."""
    from naws_hip import lib
    L = lib.load()
    p = ctypes.c_void_p
    buf = (ctypes.c_float * 64)()
    a = ctypes.cast(buf, p)
    none = p(0)
    # fp32x3 GEMM: K % 16, null operand, bad epilogue
    args = lambda k, A: (32, 32, k, A, 32 * 16, 0, a, 32 * 16, 0, a, 32, 1, 0, 0, 0,
                         lib.EPI_NONE, none, 0, none, 0, 1.0, 0.0, 0, 0, none)
    assert L.naws_gemm_f32x3_nt(*args(24, a)) == lib.ERR_ARG
    assert L.naws_gemm_f32x3_nt(*args(32, none)) == lib.ERR_NULL
    bad = list(args(32, a)); bad[15] = 99
    assert L.naws_gemm_f32x3_nt(*bad) == lib.ERR_ARG
    assert L.naws_gemm_bf16_slab_nt(32, 32, 48, a, 512, a, 512, a, 32, 1, 0, 0, 0, lib.EPI_NONE,
                                    none, 0, none, 0, 1.0, 0.0, 0, 0, none) == lib.ERR_ARG
    # fp16x2 GEMM: K % 32, missing scale vector; split: kpad = K rounded up to 32
    h2 = lambda k, sa: (32, 32, k, a, 32 * 16, 0, sa, a, 32 * 16, 0, a, a, 32, 1, 0, 0, 0, 0, 0,
                        lib.EPI_NONE, none, 0, none, 0, 1.0, 0.0, 0, 0, none)
    assert L.naws_gemm_f32_f16x2_nt(*h2(48, a)) == lib.ERR_ARG
    assert L.naws_gemm_f32_f16x2_nt(*h2(64, none)) == lib.ERR_NULL
    assert L.naws_split_f16x2(a, 1, 4, 20, 20, 0, 0, 48, a, a, none) == lib.ERR_ARG
    assert L.naws_split_f16x2(a, 1, 4, 20, 20, 0, 0, 32, a, none, none) == lib.ERR_NULL
    # split: kpad must be K rounded up to 16 (64 for the one-plane form)
    assert L.naws_split_bf16x3(a, 1, 4, 20, 20, 0, 0, 20, a, none) == lib.ERR_ARG
    assert L.naws_to_bf16_slab(a, 1, 4, 20, 20, 0, 0, 32, a, none) == lib.ERR_ARG
    assert L.naws_split_bf16x3(a, 1, 0, 20, 20, 0, 0, 32, a, none) == lib.ERR_SHAPE
    # conv: Cin % 16, relu without bias
    assert L.naws_conv3x3_nhwc_f32x3_fwd(a, a, a, 1, 8, 8, 24, 64, 1, 1, a, none) == lib.ERR_UNSUPPORTED
    assert L.naws_conv3x3_nhwc_f32x3_fwd(a, a, none, 1, 8, 8, 32, 64, 1, 1, a, none) == lib.ERR_ARG
    # NMS: too many boxes for the scan kernel, null keep
    assert L.naws_nms_sorted_fwd(a, a, 1, 20000, 0.5, a, a, none) == lib.ERR_UNSUPPORTED
    assert L.naws_nms_sorted_fwd(a, a, 1, 64, 0.5, a, none, none) == lib.ERR_NULL
    assert L.naws_nms_workspace_bytes(3, 4000) == 3 * 4000 * 63 * 8
    # image prep: crop outside the image, non-positive scale
    m3 = (ctypes.c_float * 3)(0, 0, 0)
    mp = ctypes.cast(m3, p)
    assert L.naws_prep_image_fwd(a, 8, 8, 0, 0, 0, 9, 8, mp, mp, 1.0, 0, 1.0, 1.0, 9, 8, 72, 8, a,
                                 none) == lib.ERR_SHAPE
    assert L.naws_prep_image_fwd(a, 8, 8, 0, 0, 0, 8, 8, mp, mp, 0.0, 0, 1.0, 1.0, 8, 8, 64, 8, a,
                                 none) == lib.ERR_ARG
    # MinEntropyLoss
    assert L.naws_min_entropy_loss_fwd(a, a, 0, 4, a, none) == lib.ERR_SHAPE
    assert L.naws_min_entropy_loss_bwd(a, a, none, 2, 4, a, none) == lib.ERR_NULL


def test_process_wide_state_entry_points():
    """include/naws.h lists the library's only process-wide state: the per-(kernel, device)
    launch-attribute record (resettable) and the A/B knobs (unknown knob = NAWS_ERR_ARG);
    neither needs a GPU to be exercised, and the library reads no environment variables."""
    from naws_hip import lib
    L = lib.load()
    assert L.naws_launch_state_reset() == lib.OK
    for knob in ('gemm', 'x3', 'h2', 'conv_ring', 'conv_bn', 'roi_nw', 'wino'):
        assert L.naws_set_variant(knob.encode(), 0 if knob not in ('conv_ring', 'roi_nw')
                                  else {'conv_ring': 11, 'roi_nw': 42}[knob]) == lib.OK
    assert L.naws_set_variant(b'no_such_knob', 1) == lib.ERR_ARG
    assert L.naws_set_variant(None, 1) == lib.ERR_NULL
    import subprocess
    out = subprocess.run(['nm', '-D', '--undefined-only', lib.LIB_PATH], stdout=subprocess.PIPE,
                         text=True).stdout
    assert 'getenv' not in out


def test_sgd_plane_region_table_layout():
    """naws_sgd_plane_region as the ctypes mirror fills it (no launch): a region with planes =
    None is the "updated elsewhere" marker (null planes, null vectors), a column-maxima vector is
    optional and must be int32 [rows / rows_per_batch, cols]; the ctypes struct has the layout a C
    compiler gives include/naws.h's (size and the offset of every pointer field)."""
    import ctypes as C
    import subprocess
    import tempfile
    import torch
    from naws_hip import ops
    fields = ['planes', 'plane_stride', 'bound', 'rowmax', 'inv_scale', 'colmax']
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 't.c')
        with open(src, 'w') as f:
            f.write('#include <stdio.h>\n#include <stddef.h>\n#include "naws.h"\nint main(void){'
                    'printf("%zu", sizeof(naws_sgd_plane_region));' +
                    ''.join('printf(" %%zu", offsetof(naws_sgd_plane_region, %s));' % n for n in fields) +
                    'return 0;}\n')
        exe = os.path.join(d, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), src, '-o', exe])
        want = [int(x) for x in subprocess.check_output([exe]).split()]
    got = [C.sizeof(ops._SgdPlaneRegion)] + [getattr(ops._SgdPlaneRegion, n).offset for n in fields]
    assert got == want, (got, want)
    rows, cols, rpb = 64, 256, 32
    planes = torch.zeros((2, 2, cols // 16, rpb, 16), dtype=torch.float16)
    bound = torch.zeros((rows,), dtype=torch.int32)
    rowmax = torch.zeros((rows,), dtype=torch.int32)
    inv = torch.zeros((rows,), dtype=torch.float32)
    cm = torch.zeros((rows // rpb, cols), dtype=torch.int32)
    t = ops.SgdPlaneRegions([(0, rows, cols, rows, None, None, None, None),
                             (rows * cols, rows, cols, rpb, planes, bound, rowmax, inv, cm)])
    a, b = t.host[0], t.host[1]
    assert (a.planes, a.bound, a.rowmax, a.inv_scale, a.colmax) == (None,) * 5
    assert (a.start, a.rows, a.cols, a.rows_per_batch) == (0, rows, cols, rows)
    assert b.planes == planes.data_ptr() and b.colmax == cm.data_ptr()
    assert b.plane_stride == planes.stride(0) and b.start == rows * cols
    with pytest.raises(TypeError):
        ops.SgdPlaneRegions([(0, rows, cols, rpb, planes, bound, rowmax, inv, cm[:1])])
    with pytest.raises(TypeError):
        ops.SgdPlaneRegions([(0, rows, cols, rpb, planes.float(), bound, rowmax, inv)])


def test_cu_mask_words():
    from naws_hip import ops
    assert ops.cu_mask_every(256, 8) == [0x01010101] * 8
    assert ops.cu_mask_every(64, 4, 1) == [0x22222222] * 2
    assert ops.cu_mask_every(40, 1) == [0xFFFFFFFF, 0xFF]


def test_public_header_is_plain_c_and_cpp():
    """include/naws.h compiles on its own as C and as C++ (extern "C", plain pointers and sizes: no
    torch or HIP types in the signatures)."""
    import subprocess
    hdr = os.path.join(ROOT, 'include', 'naws.h')
    for cc, lang in (('gcc', 'c'), ('g++', 'c++')):
        subprocess.check_call([cc, '-fsyntax-only', '-Wall', '-Werror', '-x', lang, hdr])
    code = re.sub(r'/\*.*?\*/', '', open(hdr).read(), flags=re.S)        # (comments cite pytorch/caffe2)
    assert 'torch' not in code and 'hipStream_t' not in code and 'at::' not in code
