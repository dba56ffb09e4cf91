"""ctypes binding of libnaws_hip.so — the C ABI declared in include/naws.h.

This is the ONLY way the host code reaches the kernels: plain pointers and
sizes, no torch types cross the boundary.  The library must be built
(`make -C na-fwebsod_amd/csrc`, or `__graft_entry__.build()`); there is no CPU
fallback: a missing library raises at import of any op.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NAWS_LIB: another build of the same ABI (tools/ab_*.py load `make AB=1`'s libnaws_hip_ab.so,
# which also carries the superseded kernel generations)
LIB_PATH = os.environ.get('NAWS_LIB') or os.path.join(os.path.dirname(_HERE), 'lib',
                                                       'libnaws_hip.so')
# host-side spelling of the library's A/B knobs (the library itself never reads the environment)
_ENV_KNOBS = {'NAWS_GEMM_VARIANT': 'gemm', 'NAWS_X3_VARIANT': 'x3', 'NAWS_H2_VARIANT': 'h2',
              'NAWS_CONV_RING': 'conv_ring', 'NAWS_CONV_BN': 'conv_bn', 'NAWS_ROI_NW': 'roi_nw',
              'NAWS_WINO_VARIANT': 'wino', 'NAWS_SPLIT': 'split', 'NAWS_SGD_WGS': 'sgd_wgs'}

OK, ERR_SHAPE, ERR_ARG, ERR_NULL, ERR_LAUNCH, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
_ERR_NAMES = {
    ERR_SHAPE: 'NAWS_ERR_SHAPE', ERR_ARG: 'NAWS_ERR_ARG', ERR_NULL: 'NAWS_ERR_NULL',
    ERR_LAUNCH: 'NAWS_ERR_LAUNCH', ERR_UNSUPPORTED: 'NAWS_ERR_UNSUPPORTED',
}
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_RELU_DROP, EPI_GATE_POS = range(5)
UN_LOG, UN_SCALE, UN_REPLACE_NAN, UN_LEAKY_RELU, UN_CLIP, UN_RELU = range(6)
BIN_ADD, BIN_SUB, BIN_MUL, BIN_DIV, BIN_GATE_POS = range(5)
PLANES_F16X2, PLANES_BF16X3, PLANES_BF16 = range(3)

p, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float

# name -> argtypes; every function returns int except the two noted below.
PROTOTYPES = {
    'naws_conv3x3_c3_nchw_to_nhwc_fwd': [p, p, p, i32, i32, i32, i32, i32, p, p],
    'naws_conv3x3_pack_weight': [p, i32, i32, p, p],
    'naws_conv3x3_nhwc_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p],
    'naws_maxpool2x2_nhwc_fwd': [p, i32, i32, i32, i32, i32, p, p],
    'naws_winograd_weight_transform': [p, i32, i32, p, p],
    'naws_conv3x3_winograd_nhwc_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, p],
    'naws_nchw_to_nhwc': [p, i32, i32, i32, i32, p, p],
    'naws_nhwc_to_nchw': [p, i32, i32, i32, i32, p, p],
    'naws_roi_pool_f_fwd': [p, i32, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, p, p],
    'naws_roi_feature_boost_fwd': [p, p, i32, i32, p, p],
    'naws_roi_feature_boost_bwd': [p, p, i32, i32, p, p],
    'naws_roi_iou_fwd': [p, i32, p, p],
    'naws_gemm_f32': [i32, i32, i32, i32, i32, p, i32, p, i32, p, i32, i32, i64, i64, i64,
                      i32, p, i64, p, i32, f32, f32, u64, i32, p],
    'naws_dropout_mask': [u64, f32, i64, p, p],
    'naws_colsum_f32': [p, i32, i32, i32, p, i32, p],
    'naws_wsddn_outputs_fwd': [p, p, p, p, i32, p, i32, i32, i32, p, p, p, p, p],
    'naws_wsddn_outputs_bwd': [p, p, p, p, p, p, i32, i32, i32, i32, p, p, p, p, i32, p],
    'naws_entropy_gate_fwd': [p, p, p, p, p, i32, i32, i32, i32, p, p, p, p, p, p],
    'naws_weighted_ce_fwd': [p, p, p, i32, i32, i32, i32, p, p],
    'naws_weighted_ce_bwd': [p, p, p, p, i32, i32, i32, i32, p, p],
    'naws_weighted_ce_shared_fwd': [p, p, p, i32, i32, i32, i32, i32, p, p],
    'naws_weighted_ce_shared_bwd': [p, p, p, p, f32, i32, i32, i32, i32, i32, p, p],
    'naws_acm_sgd_update': [p, p, p, p, p, i64, p, p, p, i32, f32, i32, i32, i32, i64, p],
    'naws_acm_sgd_update_rowmax': [p, p, p, p, p, i64, p, p, p, i32, f32, i32, i32, i32, i64, p, p,
                                   i32, p],
    'naws_acm_sgd_update_planes': [i32, p, p, p, p, i64, p, p, p, i32, f32, i32, i32, i64, p, i32, p, i32, p],
    'naws_acm_sgd_update_f16x2': [p, p, p, p, i64, p, p, p, i32, f32, i32, i32, i64, p, i32, p, i32, p],
    'naws_split_f16x2_rows_if': [p, i32, i32, i32, i32, i64, p, p, p, i32, p, i32, p],
    'naws_split_f16x2_row_range_if': [p, i32, i32, i32, i32, i32, p, p, p, i32, p, i32, p],
    'naws_roi_label_fwd': [p, p, p, p, i32, i32, i32, f32, f32, f32, i32, i32, i32, p, p, p, p, p],
    'naws_softmax_with_loss_n_fwd': [p, p, p, i32, i32, f32, p, p, p, p],
    'naws_softmax_with_loss_n_bwd': [p, p, p, p, i32, i32, f32, p, p, p],
    'naws_roi_entropy_fwd': [p, p, i32, i32, i32, p, p, i32, p],
    'naws_roi_dedup_fwd': [p, p, i32, i32, p, f32, p, p, p, p, p, p],
    'naws_tta_accumulate': [p, p, i32, i32, i32, p, p],
    'naws_tta_finish': [p, i64, i32, p],
    'naws_det_limit_fwd': [p, p, i32, i32, i32, i32, i32, p, p, p, p, p],
    'naws_stat_accumulate': [p, p, i32, i32, p, p, p],
    'naws_unary_f32': [i32, p, i64, f32, f32, p, p],
    'naws_binary_f32': [i32, p, i32, i32, p, i32, i32, p, i32, i32, p],
    'naws_softmax_rows_fwd': [p, i32, i32, p, p],
    'naws_softmax_rows_bwd': [p, p, i32, i32, p, p],
    'naws_transpose2d_f32': [p, i32, i32, p, p],
    'naws_reduce_sum_axis0': [p, i32, i32, p, p],
    'naws_gemm_bf16_nt': [i32, i32, i32, p, i32, i32, p, i32, i32, p, i32, i32, i64, i64, i64,
                          i32, p, i64, p, i32, f32, f32, u64, i32, p],
    'naws_conv3x3_nhwc_bf16_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p],
    'naws_conv3x3_nhwc_bf16_wp_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p, p],
    'naws_transpose_to_bf16': [p, i32, i32, i32, i32, i32, p, p],
    'naws_conv3x3_nhwc_f32x3_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p],
    'naws_conv3x3_nhwc_f32x3_pool_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, p, p],
    'naws_conv3x3_winograd_nhwc_f32x3_fwd': [p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, p],
    'naws_soft_nms_fwd': [p, p, i32, i32, f32, f32, f32, i32, p, p, p, p],
    'naws_nms_sorted_fwd': [p, p, i32, i32, f32, p, p, p],
    'naws_prep_image_fwd': [p, i32, i32, i32, i32, i32, i32, i32, p, p, C.c_double, i32, f32, f32, i32, i32,
                            i64, i32, p, p],
    'naws_min_entropy_loss_fwd': [p, p, i32, i32, p, p],
    'naws_min_entropy_loss_bwd': [p, p, p, i32, i32, p, p],
    'naws_to_bf16_slab': [p, i32, i32, i32, i32, i64, i32, i32, p, p],
    'naws_gemm_bf16_slab_nt_sgd': [i32, i32, i32, p, i64, p, i64, p, p, i32, p, f32, f32, f32, i32, i32,
                                   i64, p, i32, p],
    'naws_gemm_bf16_slab_nt': [i32, i32, i32, p, i64, p, i64, p, i32, i32, i64, i64, i64,
                               i32, p, i64, p, i32, f32, f32, u64, i32, p],
    'naws_split_bf16x3': [p, i32, i32, i32, i32, i64, i32, i32, p, p],
    'naws_gemm_f32x3_nt': [i32, i32, i32, p, i64, i64, p, i64, i64, p, i32, i32, i64, i64, i64,
                           i32, p, i64, p, i32, f32, f32, u64, i32, p],
    'naws_split_f16x2': [p, i32, i32, i32, i32, i64, i32, i32, p, p, p],
    'naws_split_f16x2_kscaled': [p, i32, i32, i32, i32, i64, i32, i32, p, p, p, p],
    'naws_roi_pool_f_f16x2_fwd': [p, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, i32, p, p, p],
    'naws_f16_planes_transpose': [p, i32, i32, i32, p, p],
    'naws_bf16_slab_transpose': [p, i32, i32, i32, p, p],
    'naws_roi_pool_f_bf16_slab_mapped_fwd': [p, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, p, p, p],
    'naws_roi_pool_f_nhwc_hier_fwd': [p, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, p, p],
    'naws_roi_pool_f_f16x2_hier_fwd': [p, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, i32, p, p, p,
                                       p],
    'naws_roi_maxmaps_fwd': [p, i32, i32, i32, i32, p, p, p],
    'naws_roi_pool_f_f16x2_mapped_fwd': [p, i32, i32, i32, i32, p, i32, p, i32, i32, f32, p, i32, p, p, p,
                                         p, p],
    'naws_roi_pool_f_f16x2_mapped_range_fwd': [p, i32, i32, i32, i32, p, i32, i32, i32, p, i32, i32, f32, p,
                                               i32, p, p, p, p, p],
    'naws_conv3x3_nhwc_f16x2_fwd': [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, f32, f32, p, i32,
                                    i32, p],
    'naws_amax_f32': [p, i64, p, p],
    'naws_gemm_f32_splitk': [i32, i32, i32, i32, i32, p, i32, p, i32, p, i32, i32, i64, i64, i64, i32, p,
                             i64, i32, p, p],
    'naws_launch_state_reset': [],
    'naws_set_variant': [C.c_char_p, i32],
    'naws_stream_create': [i32, p, i32, p],
    'naws_stream_destroy': [p],
    'naws_emulate_exchange': [p, p, i64, i32, f32, p],
    'naws_gemm_f32_f16x2_nt_xk': [i32, i32, i32, p, i64, i64, p, p, i64, i64, i32, p, p, i32, p],
    'naws_gemm_f32_f16x2_nt_xk_sgd': [i32, i32, i32, p, i64, i64, p, p, i64, i64, i32, p, p, p, i32, p,
                                      f32, f32, f32, i32, i32, i64, p, i64, i32, p, p, p, p, i32, p],
    'naws_conv3x3_winograd_nhwc_f16x2_fwd': [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, p, p, p],
    'naws_winograd4_weight_transform': [p, i32, i32, p, p],
    'naws_conv3x3_winograd4_nhwc_f16x2_fwd': [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, p, p, p],
    'naws_gemm_f32_f16x2_nt_amax': [i32, i32, i32, p, i64, i64, p, p, i64, i64, p, p, i32, i32,
                                    i64, i64, i64, i64, i64, i32, p, i64, p, i32, f32, f32, u64, i32,
                                    p, i32, p, p, p],
    'naws_gemm_f32_f16x2_nt_cols': [i32, i32, i32, p, i64, i64, p, p, i64, i64, p, p, i32, i32, p, f32,
                                    u64, p, i32, p, i32, i32, p],
    'naws_gemm_f32_amax': [i32, i32, i32, i32, i32, p, i32, p, i32, p, i32, i32, i64, i64, i64,
                           i32, p, i64, p, i32, f32, f32, u64, i32, p, i32, p, p, p],
    'naws_split_f16x2_dual': [p, i32, i32, i32, i32, i64, p, p, p, p, p, i32, p, p, i32, p],
    'naws_gemm_f32_f16x2_nt': [i32, i32, i32, p, i64, i64, p, p, i64, i64, p, p, i32, i32,
                               i64, i64, i64, i64, i64, i32, p, i64, p, i32, f32, f32, u64, i32, p],
}
SPECIAL = {
    'naws_version': ([], C.c_char_p),
    'naws_last_hip_error': ([], i32),
    'naws_entropy_gate_workspace_floats': ([i32, i32, i32, i32], i64),
    'naws_winograd_workspace_floats': ([i32, i32, i32, i32, i32, i32], i64),
    'naws_nms_workspace_bytes': ([i32, i32], i64),
    'naws_roi_label_workspace_bytes': ([i32, i32, i32], i64),
    'naws_softmax_with_loss_n_workspace_floats': ([i32], i64),
    'naws_roi_pool_workspace_floats': ([i32, i32, i32, i32], i64),
    'naws_winograd_f32x3_workspace_floats': ([i32, i32, i32, i32, i32, i32], i64),
    'naws_winograd_f16x2_workspace_floats': ([i32, i32, i32, i32, i32, i32], i64),
    'naws_winograd4_f16x2_workspace_floats': ([i32, i32, i32, i32, i32, i32], i64),
    'naws_gemm_f32_splitk_workspace_floats': ([i32, i32, i32, i32], i64),
}
ALL_SYMBOLS = sorted(list(PROTOTYPES) + list(SPECIAL))
# entries that exist only in the A/B build (make AB=1, loaded through NAWS_LIB): experiments whose
# losing arm is kept reproducible for the tools, never part of the product ABI
AB_PROTOTYPES = {
    'naws_conv3x3_winograd_nhwc_f16x2_col_fwd': [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p, p, p,
                                                 p, p],
}


class NawsError(RuntimeError):
    """Raised where the reference op's CAFFE_ENFORCE would throw."""

    def __init__(self, fn, code, hip_error=0):
        self.fn, self.code, self.hip_error = fn, code, hip_error
        msg = '%s failed: %s' % (fn, _ERR_NAMES.get(code, code))
        if code == ERR_LAUNCH:
            msg += ' (hipError %d)' % hip_error
        super(NawsError, self).__init__(msg)


_lib = None


def load():
    """Load the library once; loud failure if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch's bundled HIP runtime has to be in the process BEFORE this library is opened: opened
    # first, libnaws_hip.so binds /opt/rocm's libamdhip64 and a process that then touches the GPU
    # through torch holds two runtimes (the library's first launch fails with hipErrorNoDevice)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libnaws_hip.so not found at %s — build it with `make -C na-fwebsod_amd/csrc` '
            '(or __graft_entry__.build()); there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, i32
    for name, (argtypes, restype) in SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    for name, argtypes in AB_PROTOTYPES.items():       # present in libnaws_hip_ab.so only
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, i32
    _lib = lib
    for env, knob in _ENV_KNOBS.items():
        if os.environ.get(env):
            set_variant(knob, int(os.environ[env]))
    return lib


def set_variant(knob, value):
    """naws_set_variant: choose a kernel form for an A/B run (results never change)."""
    return call('naws_set_variant', knob.encode(), int(value))


def call(name, *args):
    """Invoke an int-returning entry point; raise NawsError on a negative code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != OK:
        raise NawsError(name, rc, lib.naws_last_hip_error() if rc == ERR_LAUNCH else 0)
    return rc
