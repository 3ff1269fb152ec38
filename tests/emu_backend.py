"""numpy front end of tests/emu/libaae_emu.so: the product kernel + launch sources
compiled against the CPU fiber emulator (TEST INFRASTRUCTURE ONLY, see
tests/emu/hip_emu.h).  Exposes the C ABI of include/aae_hip.h on host arrays."""
import ctypes
import os
import subprocess

import numpy as np

from augmentedautoencoder_amd import _lib
from augmentedautoencoder_amd.weights import EncoderConfig, as_pointer_array, ordered_weight_arrays

HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = None


def lib():
    global _EMU
    if _EMU is None:
        import fcntl
        with open(os.path.join(HERE, 'emu', '.build.lock'), 'w') as lock:      # pytest -n: one worker builds, the others wait
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(['make', '-s', '-C', os.path.join(HERE, 'emu')])
        _EMU = _lib.declare(ctypes.CDLL(os.path.join(HERE, 'emu', 'libaae_emu.so')))
    return _EMU


def set_block_order(order):
    """0: blocks of a grid run in ascending order, 1: descending, 2: a scrambled permutation (kernels that let the
    last block to arrive finish a reduction must give the same bits in every order)."""
    lib().aae_emu_set_block_order(int(order))


def _aligned(nbytes, align=256):
    # workspaces arrive with arbitrary contents (the ticket words of block_ticket_arrive included): never zeros
    raw = np.full(nbytes + align, 0xA5, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


def split_k_small_batches(enc):
    """Switch an encoder to the 128 x 128 split-K igemm + separate reduce launches for small batches too (the kernels
    the wave-split-K / ticketed path replaced as the default; kept as options, so their tests keep running)."""
    for name in ('wavek', 'wavek_dense', 'gemv_ticket', 'first_group_split_max_tiles'):     # (conv1: four pixel groups per block, too)
        enc.set_option(name, 0)
    return enc


class EmuEncoder(object):
    def __init__(self, weights, cfg):
        self.cfg = cfg
        self.L = lib()
        self._arrays = ordered_weight_arrays(weights, cfg)
        h = ctypes.c_void_p()
        desc = cfg.to_desc()
        rc = self.L.aae_encoder_create(ctypes.byref(desc), as_pointer_array(self._arrays), len(self._arrays), ctypes.byref(h))
        _lib.check(self.L, rc, 'aae_encoder_create')
        self.h = h
        self.ws = None
        self.options = {}
        # the planner's round size is one block per compute unit of the DEVICE; the emulated device has three (hip_emu.h), the launch
        # plans under test are the MI355X's: pin its 256
        self.set_option('wavek_target_blocks', 256)

    def set_option(self, name, value):
        _lib.check(self.L, self.L.aae_encoder_set_option(self.h, name.encode(), int(value)), 'set_option')
        self.options[name] = int(value)

    def forward(self, x):
        x = np.ascontiguousarray(x)
        B = x.shape[0]
        dt = _lib.AAE_DTYPE_U8 if x.dtype == np.uint8 else _lib.AAE_DTYPE_F32
        if dt == _lib.AAE_DTYPE_F32:
            x = x.astype(np.float32)
        n = self.L.aae_encoder_workspace_bytes(self.h, B)
        self.ws = _aligned(n)
        z = np.zeros((B, self.cfg.latent_space_size), dtype=np.float32)
        rc = self.L.aae_encoder_forward(self.h, x.ctypes.data, dt, B, z.ctypes.data, self.ws.ctypes.data, n, None)
        _lib.check(self.L, rc, 'aae_encoder_forward')
        self.B = B
        return z

    def activation(self, layer):
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self.L, self.L.aae_encoder_activation_info(self.h, self.B, layer, ctypes.byref(off), ctypes.byref(cnt)), 'info')
        H, W, Ci, Ho, Wo, Co = self.cfg.layer_shapes()[layer]
        raw = self.ws[off.value:off.value + 4 * cnt.value]
        if self.L.aae_encoder_split_precision_for_batch(self.h, int(self.B)):      # f32x3h: fp16 (hi, lo) pairs of value * 2^shift, [pixel][chunk][hi x 32 | lo x 32]
            pairs = raw.view(np.float16).reshape(-1, 2, 32).astype(np.float64)          # chunks of the flat [B*Ho*Wo*Co] index
            return ((pairs[:, 0, :] + pairs[:, 1, :]).reshape(self.B, Ho, Wo, Co)
                    / 2.0 ** self.options.get('x3h_act_shift', 4)).astype(np.float32)
        return raw.view(np.float32).reshape(self.B, Ho, Wo, Co).copy()

    def labels(self):
        out, i = [], 0
        while True:
            s = self.L.aae_encoder_kernel_label(self.h, i)
            if not s:
                return out
            out.append(s.decode())
            i += 1

    def close(self):
        if self.h:
            self.L.aae_encoder_destroy(self.h)
            self.h = None


class EmuCodebook(object):
    def __init__(self, E, dtype='f32'):
        self.L = lib()
        self.E = np.ascontiguousarray(E, dtype=np.float32)
        h = ctypes.c_void_p()
        if dtype == 'bf16':
            from augmentedautoencoder_amd.weights import to_bf16_bits
            bits = to_bf16_bits(self.E)
            rc = self.L.aae_codebook_create(bits.ctypes.data, self.E.shape[0], self.E.shape[1], _lib.AAE_DTYPE_BF16, 0, ctypes.byref(h))
        else:
            rc = self.L.aae_codebook_create(self.E.ctypes.data, self.E.shape[0], self.E.shape[1], _lib.AAE_DTYPE_F32, 0, ctypes.byref(h))
        _lib.check(self.L, rc, 'aae_codebook_create')
        self.h = h

    def set_mode(self, mode):
        _lib.check(self.L, self.L.aae_codebook_set_scan_mode(self.h, mode), 'set_scan_mode')

    def prepare_upright(self, col_stride):
        _lib.check(self.L, self.L.aae_codebook_prepare_upright(self.h, int(col_stride), None), 'prepare_upright')

    def nn(self, z, topk=1, col_stride=1):
        z = np.ascontiguousarray(z, dtype=np.float32)
        B = z.shape[0]
        n = self.L.aae_codebook_workspace_bytes(self.h, B, topk)
        ws = _aligned(n)
        idx = np.full((B, topk), -7, dtype=np.int64)
        score = np.zeros((B, topk), dtype=np.float32)
        rc = self.L.aae_codebook_nn(self.h, z.ctypes.data, B, topk, col_stride, idx.ctypes.data, score.ctypes.data,
                                    ws.ctypes.data, n, None)
        _lib.check(self.L, rc, 'aae_codebook_nn')
        return idx, score

    def similarity(self, z):
        z = np.ascontiguousarray(z, dtype=np.float32)
        B = z.shape[0]
        n = self.L.aae_codebook_workspace_bytes(self.h, B, 1)
        ws = _aligned(n)
        cs = np.full((B, self.E.shape[0]), np.nan, dtype=np.float32)
        rc = self.L.aae_codebook_similarity(self.h, z.ctypes.data, B, cs.ctypes.data, ws.ctypes.data, n, None)
        _lib.check(self.L, rc, 'aae_codebook_similarity')
        return cs

    def close(self):
        if self.h:
            self.L.aae_codebook_destroy(self.h)
            self.h = None


def encode_nn(enc, cb, x, col_stride=1):
    """aae_encode_nn on the emulator: (z, idx [B,1], score [B,1]) -- the fused per-detection call, where the first
    encoder kernel prepares the ticket words of the later launches (the scan's included)."""
    L = lib()
    x = np.ascontiguousarray(x)
    B = x.shape[0]
    dt = _lib.AAE_DTYPE_U8 if x.dtype == np.uint8 else _lib.AAE_DTYPE_F32
    if dt == _lib.AAE_DTYPE_F32:
        x = x.astype(np.float32)
    n_e = L.aae_encoder_workspace_bytes(enc.h, B)
    n_c = L.aae_codebook_workspace_bytes(cb.h, B, 1)
    enc.ws, ws_c = _aligned(n_e), _aligned(n_c)
    enc.B = B
    z = np.zeros((B, enc.cfg.latent_space_size), dtype=np.float32)
    idx = np.full((B, 1), -7, dtype=np.int64)
    score = np.zeros((B, 1), dtype=np.float32)
    rc = L.aae_encode_nn(enc.h, cb.h, x.ctypes.data, dt, B, int(col_stride), z.ctypes.data, idx.ctypes.data, score.ctypes.data,
                         enc.ws.ctypes.data, n_e, ws_c.ctypes.data, n_c, None)
    _lib.check(L, rc, 'aae_encode_nn')
    return z, idx, score


def crop_resize(img, boxes_xywh_size, out_hw):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    boxes = np.ascontiguousarray(np.asarray(boxes_xywh_size, dtype=np.int32).reshape(-1, 5))
    out = np.full((boxes.shape[0], out_hw[0], out_hw[1], img.shape[2]), 123, dtype=np.uint8)
    rc = L.aae_crop_resize_u8(img.ctypes.data, img.shape[0], img.shape[1], img.shape[2], boxes.ctypes.data, boxes.shape[0],
                              out_hw[0], out_hw[1], out.ctypes.data, None)
    _lib.check(L, rc, 'aae_crop_resize_u8')
    return out


def l2_normalize(z):
    L = lib()
    z = np.ascontiguousarray(z, dtype=np.float32)
    q = np.zeros_like(z)
    _lib.check(L, L.aae_l2_normalize(z.ctypes.data, z.shape[0], z.shape[1], q.ctypes.data, None), 'aae_l2_normalize')
    return q


class EmuDecoder(object):
    def __init__(self, weights, cfg):
        from augmentedautoencoder_amd.weights import ordered_decoder_weight_arrays
        self.cfg = cfg
        self.L = lib()
        self._arrays = ordered_decoder_weight_arrays(weights, cfg)
        h = ctypes.c_void_p()
        desc = cfg.to_desc()
        rc = self.L.aae_decoder_create(ctypes.byref(desc), as_pointer_array(self._arrays), len(self._arrays), ctypes.byref(h))
        _lib.check(self.L, rc, 'aae_decoder_create')
        self.h = h

    def forward(self, z):
        z = np.ascontiguousarray(z, dtype=np.float32)
        B = z.shape[0]
        n = self.L.aae_decoder_workspace_bytes(self.h, B)
        self.ws = _aligned(n)
        out = np.full((B,) + tuple(self.cfg.shape), np.nan, dtype=np.float32)
        rc = self.L.aae_decoder_forward(self.h, z.ctypes.data, B, out.ctypes.data, self.ws.ctypes.data, n, None)
        _lib.check(self.L, rc, 'aae_decoder_forward')
        self.B = B
        return out

    def activation(self, stage):
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self.L, self.L.aae_decoder_activation_info(self.h, self.B, stage, ctypes.byref(off), ctypes.byref(cnt)), 'info')
        dims = self.cfg.layer_dimensions()[stage]
        return self.ws[off.value:off.value + 4 * cnt.value].view(np.float32).reshape(self.B, dims[0], dims[1],
                                                                                      self.cfg.num_filters[stage]).copy()

    def labels(self):
        out, i = [], 0
        while True:
            s = self.L.aae_decoder_kernel_label(self.h, i)
            if not s:
                return out
            out.append(s.decode())
            i += 1

    def close(self):
        if self.h:
            self.L.aae_decoder_destroy(self.h)
            self.h = None


# ---- grouped multi-object query (aae_encode_nn_multi / aae_codebook_nn_multi / aae_detect_nn_multi) ----------------------------
def _multi_items(items):
    arr = (_lib.MultiItem * len(items))()
    for k, (enc, cb, n, stride) in enumerate(items):
        arr[k].enc = enc.h if enc is not None else None
        arr[k].cb = cb.h
        arr[k].n = int(n)
        arr[k].col_stride = int(stride)
    return arr


class MultiWorkspace(object):
    """One grow-only scratch buffer reused from call to call (what an estimator does): whatever the previous frame left in
    it -- activations, partials, ticket words of OTHER objects at other offsets -- is what the next frame finds."""

    def __init__(self):
        self.buf = _aligned(256)

    def get(self, n):
        if len(self.buf) < n:
            self.buf = _aligned(n)
        return self.buf


def encode_nn_multi(items, x, ws=None):
    """items: [(EmuEncoder, EmuCodebook, n, col_stride)]; x: the crops of all items, concatenated in item order.
    Returns (z [rows,J], idx [rows], score [rows], launches of the grouped part)."""
    L = lib()
    arr = _multi_items(items)
    x = np.ascontiguousarray(x)
    dt = _lib.AAE_DTYPE_U8 if x.dtype == np.uint8 else _lib.AAE_DTYPE_F32
    rows = L.aae_multi_rows(arr, len(items))
    assert rows == x.shape[0], (rows, x.shape)
    nbytes = L.aae_multi_workspace_bytes(arr, len(items), 0)
    buf = (ws or MultiWorkspace()).get(nbytes)
    J = items[0][0].cfg.latent_space_size
    z = np.full((rows, J), np.nan, dtype=np.float32)
    idx = np.full((rows,), -7, dtype=np.int64)
    score = np.full((rows,), np.nan, dtype=np.float32)
    rc = L.aae_encode_nn_multi(arr, len(items), x.ctypes.data, dt, z.ctypes.data, idx.ctypes.data, score.ctypes.data, buf.ctypes.data, nbytes, None)
    _lib.check(L, rc, 'aae_encode_nn_multi')
    return z, idx, score, L.aae_multi_last_launches()


def codebook_nn_multi(items, z, ws=None):
    L = lib()
    arr = _multi_items(items)
    z = np.ascontiguousarray(z, dtype=np.float32)
    rows = L.aae_multi_rows(arr, len(items))
    assert rows == z.shape[0]
    nbytes = L.aae_multi_workspace_bytes(arr, len(items), 1)
    buf = (ws or MultiWorkspace()).get(nbytes)
    idx = np.full((rows,), -7, dtype=np.int64)
    score = np.full((rows,), np.nan, dtype=np.float32)
    rc = L.aae_codebook_nn_multi(arr, len(items), z.ctypes.data, idx.ctypes.data, score.ctypes.data, buf.ctypes.data, nbytes, None)
    _lib.check(L, rc, 'aae_codebook_nn_multi')
    return idx, score, L.aae_multi_last_launches()


def detect_nn_multi(items, img, boxes, ws=None):
    L = lib()
    arr = _multi_items(items)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 5))
    rows = L.aae_multi_rows(arr, len(items))
    assert rows == boxes.shape[0]
    cfg = items[0][0].cfg
    crops = np.full((rows,) + tuple(cfg.shape), 123, dtype=np.uint8)
    nbytes = L.aae_multi_workspace_bytes(arr, len(items), 0)
    buf = (ws or MultiWorkspace()).get(nbytes)
    z = np.full((rows, cfg.latent_space_size), np.nan, dtype=np.float32)
    idx = np.full((rows,), -7, dtype=np.int64)
    score = np.full((rows,), np.nan, dtype=np.float32)
    rc = L.aae_detect_nn_multi(arr, len(items), img.ctypes.data, img.shape[0], img.shape[1], img.shape[2], boxes.ctypes.data, crops.ctypes.data,
                               z.ctypes.data, idx.ctypes.data, score.ctypes.data, buf.ctypes.data, nbytes, None)
    _lib.check(L, rc, 'aae_detect_nn_multi')
    return crops, z, idx, score
