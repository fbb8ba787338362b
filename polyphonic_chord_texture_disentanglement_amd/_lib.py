"""ctypes binding of libptvae_hip.so (C ABI in include/ptvae_hip.h).

The library is the product: there is NO fallback.  If it is missing, or a call returns a
non-zero status, this module raises -- loudly.  `python __graft_entry__.py` (or
`make -C polyphonic_chord_texture_disentanglement_amd/csrc`) builds it in-tree.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libptvae_hip.so')

PREC_F32, PREC_BF16 = 0, 1
_PREC = {'fp32': PREC_F32, 'f32': PREC_F32, 'bf16': PREC_BF16, 0: 0, 1: 1}

_lib = None

c_f = ctypes.c_void_p       # float* (device)
c_i = ctypes.c_int
c_l = ctypes.c_long
c_fl = ctypes.c_float

_SIGNATURES = {
    'ptv_arch': (ctypes.c_char_p, []),
    'ptv_abi_version': (c_i, []),
    'ptv_gemm': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_l, c_f, c_fl,
                       c_i, c_i, c_i, c_f]),
    'ptv_gru_seq_fwd': (c_i, [c_i, c_i, c_i, c_i, c_f, c_l, c_l, c_f, c_l, c_l, c_f, c_f, c_f, c_f,
                              c_f, c_i, c_f]),
    'ptv_gru_seq_bwd': (c_i, [c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_l, c_l, c_f, c_l,
                              c_f, c_f, c_f, c_f, c_i, c_f]),
}


def exported_symbols():
    """Names every entry point declared in include/ptvae_hip.h (checked by the CPU tests)."""
    return list(_SIGNATURES.keys())


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libptvae_hip.so not found at %s -- the HIP extension is the product path and '
                'there is no CPU fallback. Build it: python -c "import __graft_entry__ as g; '
                'g.build()"' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def prec_code(p):
    return _PREC[p]


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8), \
        'device fp32/int tensor expected, got %s %s' % (t.device, t.dtype)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError('libptvae_hip: %s failed with status %d' % (what, rc))


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    check(rc, name)
