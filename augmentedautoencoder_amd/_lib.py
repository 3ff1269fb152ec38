"""ctypes binding of libaae_hip.so (include/aae_hip.h).

The product path has exactly one back end: the hand-written HIP library built
in-tree by ``__graft_entry__.build()``.  There is no CPU fallback -- if the
library or a GPU is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

AAE_OK = 0
AAE_DTYPE_U8 = 0
AAE_DTYPE_F32 = 1
AAE_DTYPE_BF16 = 2
AAE_MAX_LAYERS = 8
AAE_SCAN_AUTO, AAE_SCAN_GEMV, AAE_SCAN_MFMA, AAE_SCAN_STREAM, AAE_SCAN_STREAM_2L, AAE_SCAN_AUTO_NO_PRUNE, AAE_SCAN_STREAM_WALK = 0, 1, 2, 3, 4, 5, 6
AAE_SCAN_AUTO_PACKED, AAE_SCAN_AUTO_RH2, AAE_SCAN_AUTO_FIN = 7, 8, 9
AAE_ABI_VERSION = 3

LIB_NAME = 'libaae_hip.so'
# the same sources with -DAAE_EXPERIMENTS: every kernel variant that measured slower than the defaults + the profiling / ablation
# aids (tools/, the A/B tests).  Chosen for the whole process by AAE_EXPERIMENTS=1 in the environment; never the default.
EXPERIMENTS_LIB_NAME = 'libaae_hip_experiments.so'


def experiments_requested():
    return os.environ.get('AAE_EXPERIMENTS', '0') not in ('', '0')

# every symbol include/aae_hip.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = (
    'aae_abi_version', 'aae_has_experiments', 'aae_last_error',
    'aae_encoder_create', 'aae_encoder_destroy', 'aae_encoder_set_option', 'aae_encoder_workspace_bytes', 'aae_encoder_forward',
    'aae_encoder_forward_timed', 'aae_encoder_kernel_label', 'aae_encoder_kernel_flops',
    'aae_encoder_activation_info', 'aae_encoder_debug_timeline', 'aae_encoder_x3h_saturated', 'aae_encoder_x3h_last_slot', 'aae_encoder_x3h_poll',
    'aae_encoder_x3h_release_slot',
    'aae_encoder_split_precision_for_batch',
    'aae_codebook_create', 'aae_codebook_update', 'aae_codebook_destroy', 'aae_codebook_set_scan_mode',
    'aae_codebook_prepare_upright',
    'aae_codebook_workspace_bytes', 'aae_codebook_nn', 'aae_codebook_nn_timed', 'aae_encode_nn', 'aae_detect_nn', 'aae_codebook_similarity', 'aae_l2_normalize',
    'aae_crop_resize_u8', 'aae_pack_pairs', 'aae_unpack_pairs',
    'aae_multi_workspace_bytes', 'aae_multi_rows', 'aae_encode_nn_multi', 'aae_codebook_nn_multi', 'aae_detect_nn_multi', 'aae_multi_last_launches',
    'aae_decoder_create', 'aae_decoder_destroy', 'aae_decoder_workspace_bytes', 'aae_decoder_forward',
    'aae_decoder_forward_timed', 'aae_decoder_kernel_label', 'aae_decoder_kernel_flops', 'aae_decoder_activation_info',
)


class EncoderDesc(Structure):
    _fields_ = [
        ('in_h', c_int32), ('in_w', c_int32), ('in_c', c_int32),
        ('num_layers', c_int32),
        ('num_filters', c_int32 * AAE_MAX_LAYERS),
        ('strides', c_int32 * AAE_MAX_LAYERS),
        ('kernel_size', c_int32),
        ('latent_size', c_int32),
        ('batch_norm', c_int32),
        ('bn_eps', c_float),
    ]


class DecoderDesc(Structure):
    _fields_ = [
        ('out_h', c_int32), ('out_w', c_int32), ('out_c', c_int32),
        ('num_layers', c_int32),
        ('num_filters', c_int32 * AAE_MAX_LAYERS),
        ('strides', c_int32 * AAE_MAX_LAYERS),
        ('kernel_size', c_int32),
        ('latent_size', c_int32),
        ('batch_norm', c_int32),
        ('bn_eps', c_float),
    ]


class MultiItem(Structure):
    """aae_multi_item: (encoder handle, codebook handle, detections of that object in the frame, col_stride)"""
    _fields_ = [('enc', c_void_p), ('cb', c_void_p), ('n', c_int32), ('col_stride', c_int32)]


def declare(lib):
    """Attach argtypes/restypes for every entry point of include/aae_hip.h."""
    lib.aae_abi_version.restype = c_int
    lib.aae_abi_version.argtypes = []
    lib.aae_has_experiments.restype = c_int
    lib.aae_has_experiments.argtypes = []
    lib.aae_last_error.restype = c_char_p
    lib.aae_last_error.argtypes = []

    lib.aae_encoder_create.restype = c_int
    lib.aae_encoder_create.argtypes = [POINTER(EncoderDesc), POINTER(c_void_p), c_int, POINTER(c_void_p)]
    lib.aae_encoder_destroy.restype = None
    lib.aae_encoder_destroy.argtypes = [c_void_p]
    lib.aae_encoder_set_option.restype = c_int
    lib.aae_encoder_set_option.argtypes = [c_void_p, c_char_p, c_int]
    lib.aae_encoder_workspace_bytes.restype = c_size_t
    lib.aae_encoder_workspace_bytes.argtypes = [c_void_p, c_int]
    lib.aae_encoder_forward.restype = c_int
    lib.aae_encoder_forward.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_encoder_forward_timed.restype = c_int
    lib.aae_encoder_forward_timed.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                              POINTER(c_float), c_int, POINTER(c_int)]
    lib.aae_encoder_kernel_label.restype = c_char_p
    lib.aae_encoder_kernel_label.argtypes = [c_void_p, c_int]
    lib.aae_encoder_kernel_flops.restype = c_double
    lib.aae_encoder_kernel_flops.argtypes = [c_void_p, c_int]
    lib.aae_encoder_split_precision_for_batch.restype = c_int
    lib.aae_encoder_split_precision_for_batch.argtypes = [c_void_p, c_int]
    lib.aae_encoder_x3h_saturated.restype = c_int
    lib.aae_encoder_x3h_saturated.argtypes = [c_void_p, POINTER(c_int), c_void_p]
    lib.aae_encoder_x3h_last_slot.restype = c_int
    lib.aae_encoder_x3h_last_slot.argtypes = []
    lib.aae_encoder_x3h_poll.restype = c_int
    lib.aae_encoder_x3h_poll.argtypes = [c_void_p, POINTER(c_int), c_int, POINTER(c_int), c_void_p]
    lib.aae_encoder_x3h_release_slot.restype = c_int
    lib.aae_encoder_x3h_release_slot.argtypes = [c_void_p, c_int, c_void_p]
    lib.aae_encoder_debug_timeline.restype = c_int
    lib.aae_encoder_debug_timeline.argtypes = [c_void_p, POINTER(c_int64)]
    lib.aae_encoder_activation_info.restype = c_int
    lib.aae_encoder_activation_info.argtypes = [c_void_p, c_int, c_int, POINTER(c_size_t), POINTER(c_size_t)]

    lib.aae_codebook_create.restype = c_int
    lib.aae_codebook_create.argtypes = [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_void_p)]
    lib.aae_codebook_update.restype = c_int
    lib.aae_codebook_update.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    lib.aae_codebook_destroy.restype = None
    lib.aae_codebook_destroy.argtypes = [c_void_p]
    lib.aae_codebook_set_scan_mode.restype = c_int
    lib.aae_codebook_set_scan_mode.argtypes = [c_void_p, c_int]
    lib.aae_codebook_prepare_upright.restype = c_int
    lib.aae_codebook_prepare_upright.argtypes = [c_void_p, c_int, c_void_p]
    lib.aae_codebook_workspace_bytes.restype = c_size_t
    lib.aae_codebook_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.aae_pack_pairs.restype = c_int
    lib.aae_pack_pairs.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.aae_unpack_pairs.restype = c_int
    lib.aae_unpack_pairs.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.aae_encode_nn.restype = c_int
    lib.aae_encode_nn.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p, c_size_t, c_void_p]
    lib.aae_codebook_nn.restype = c_int
    lib.aae_codebook_nn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_codebook_nn_timed.restype = c_int
    lib.aae_codebook_nn_timed.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, POINTER(c_float)]
    lib.aae_detect_nn.restype = c_int
    lib.aae_detect_nn.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
    lib.aae_codebook_similarity.restype = c_int
    lib.aae_codebook_similarity.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_l2_normalize.restype = c_int
    lib.aae_l2_normalize.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.aae_crop_resize_u8.restype = c_int
    lib.aae_crop_resize_u8.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.aae_multi_workspace_bytes.restype = c_size_t
    lib.aae_multi_workspace_bytes.argtypes = [POINTER(MultiItem), c_int, c_int]
    lib.aae_multi_rows.restype = c_int
    lib.aae_multi_rows.argtypes = [POINTER(MultiItem), c_int]
    lib.aae_encode_nn_multi.restype = c_int
    lib.aae_encode_nn_multi.argtypes = [POINTER(MultiItem), c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_codebook_nn_multi.restype = c_int
    lib.aae_codebook_nn_multi.argtypes = [POINTER(MultiItem), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_detect_nn_multi.restype = c_int
    lib.aae_detect_nn_multi.argtypes = [POINTER(MultiItem), c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]
    lib.aae_multi_last_launches.restype = c_int
    lib.aae_multi_last_launches.argtypes = []

    lib.aae_decoder_create.restype = c_int
    lib.aae_decoder_create.argtypes = [POINTER(DecoderDesc), POINTER(c_void_p), c_int, POINTER(c_void_p)]
    lib.aae_decoder_destroy.restype = None
    lib.aae_decoder_destroy.argtypes = [c_void_p]
    lib.aae_decoder_workspace_bytes.restype = c_size_t
    lib.aae_decoder_workspace_bytes.argtypes = [c_void_p, c_int]
    lib.aae_decoder_forward.restype = c_int
    lib.aae_decoder_forward.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.aae_decoder_forward_timed.restype = c_int
    lib.aae_decoder_forward_timed.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                              POINTER(c_float), c_int, POINTER(c_int)]
    lib.aae_decoder_kernel_label.restype = c_char_p
    lib.aae_decoder_kernel_label.argtypes = [c_void_p, c_int]
    lib.aae_decoder_kernel_flops.restype = c_double
    lib.aae_decoder_kernel_flops.argtypes = [c_void_p, c_int]
    lib.aae_decoder_activation_info.restype = c_int
    lib.aae_decoder_activation_info.argtypes = [c_void_p, c_int, c_int, POINTER(c_size_t), POINTER(c_size_t)]
    return lib


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), EXPERIMENTS_LIB_NAME if experiments_requested() else LIB_NAME)


_LIB = None


def load():
    """Load libaae_hip.so.  torch is imported first so that the library binds to the
    HIP runtime torch already loaded (one runtime per process: device pointers,
    streams and events are shared between torch tensors and our kernels)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (must precede the dlopen below)
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            '%s not built: run `python -c "import __graft_entry__ as g; g.build(%s)"` at the repo root '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % (path, 'experiments=True' if experiments_requested() else ''))
    lib = declare(ctypes.CDLL(path))
    if lib.aae_abi_version() != AAE_ABI_VERSION:
        raise RuntimeError('libaae_hip.so ABI version %d != expected %d' % (lib.aae_abi_version(), AAE_ABI_VERSION))
    _LIB = lib
    return lib


def check(lib, rc, what):
    if rc != AAE_OK:
        msg = lib.aae_last_error()
        msg = msg.decode('utf-8', 'replace') if msg else ''
        if rc in (-1, -2, -4):
            raise ValueError('%s failed (%d): %s' % (what, rc, msg))
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))
