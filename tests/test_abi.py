"""The C ABI: header, binding table and built library agree; error behaviour (codes +
aae_last_error) is checked through the CPU-emulated build of the same host sources."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from augmentedautoencoder_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_table_list_the_same_symbols():
    text = open(os.path.join(ROOT, 'include', 'aae_hip.h')).read() + open(os.path.join(ROOT, 'include', 'aae_hip_tuning.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    declared = set(re.findall(r'\b(aae_[a-z0-9_]+)\s*\(', text))
    assert declared == set(_lib.EXPORTED_SYMBOLS)


def test_built_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    out = subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--dyn-syms', '--wide', g.LIB]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if ' FUNC ' in l and ' UND ' not in l)
    assert set(_lib.EXPORTED_SYMBOLS) <= exported
    # the device code object for gfx950 is embedded
    assert b'gfx950' in open(g.LIB, 'rb').read()


def test_product_loader_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from augmentedautoencoder_amd.engine import EncoderEngine
    from augmentedautoencoder_amd.weights import EncoderConfig
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        EncoderEngine(EncoderConfig(), {})


def test_error_codes_and_messages():
    import emu_backend as eb
    L = eb.lib()
    assert L.aae_abi_version() == _lib.AAE_ABI_VERSION
    h = ctypes.c_void_p()
    assert L.aae_encoder_create(None, None, 0, ctypes.byref(h)) == -1
    assert b'null' in L.aae_last_error()
    E = np.zeros((10, 6), dtype=np.float32)                          # J % 4 != 0
    assert L.aae_codebook_create(E.ctypes.data, 10, 6, _lib.AAE_DTYPE_F32, 0, ctypes.byref(h)) == -2
    assert b'latent size' in L.aae_last_error()
    E = np.zeros((10, 8), dtype=np.float32)
    assert L.aae_codebook_create(E.ctypes.data, 10, 8, 7, 0, ctypes.byref(h)) == -2          # unknown dtype
    cb = eb.EmuCodebook(np.eye(8, dtype=np.float32))
    z = np.ones((2, 8), dtype=np.float32)
    idx = np.zeros((2, 1), dtype=np.int64)
    sc = np.zeros((2, 1), dtype=np.float32)
    ws = np.zeros(64, dtype=np.uint8)
    rc = L.aae_codebook_nn(cb.h, z.ctypes.data, 2, 1, 1, idx.ctypes.data, sc.ctypes.data, ws.ctypes.data, 64, None)
    assert rc == -4 and b'workspace' in L.aae_last_error()
    with pytest.raises(ValueError):
        cb.nn(z, topk=2, col_stride=36)                              # upright is a top-1 notion (codebook.py:65-66)
    with pytest.raises(ValueError):
        cb.nn(z, topk=9)                                             # k > N
    cb.close()
