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
# the C ABI this Python package was written against (csrc/version.hip): the .so is a built artefact that ships beside the sources, and a
# stale one would load without error and silently change argument contracts (round-4 advice: counts[3] of ptv_pianotree_targets)
EXPECTED_ABI = 6

HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'ptvae_hip.h')
# instrumentation / test aids (ptv_prof_*, ptv_debug_*): bound too, but not part of the product ABI
DEBUG_HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'ptvae_hip_debug.h')


def _parse_header(path):
    """Build the ctypes signature table from include/ptvae_hip.h (the single source of truth)."""
    import re
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    sigs = {}
    for m in re.finditer(r'(const char\*|int|long)\s+(ptv_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                elif a.startswith('double '):
                    argtypes.append(ctypes.c_double)
                elif a.startswith('unsigned long long '):
                    argtypes.append(ctypes.c_ulonglong)
                elif a.startswith('long '):
                    argtypes.append(ctypes.c_long)
                elif a.startswith('unsigned '):
                    argtypes.append(ctypes.c_uint)
                elif a.startswith('int '):
                    argtypes.append(ctypes.c_int)
                elif a.startswith('float '):
                    argtypes.append(ctypes.c_float)
                else:
                    raise RuntimeError('unparsed argument %r of %s' % (a, name))
        sigs[name] = (ctypes.c_char_p if ret.startswith('const char') else (ctypes.c_long if ret == 'long' else ctypes.c_int), argtypes)
    return sigs


def header_enum(name, path=None):
    """{enumerator: value} of a plain (0, 1, 2, ...) enum of include/ptvae_hip.h: the slot numbers of a composite entry point's tables"""
    import re
    src = re.sub(r'/\*.*?\*/', '', open(path or HEADER_PATH).read(), flags=re.S)
    m = re.search(r'enum\s+%s\s*\{(.*?)\}' % name, src, flags=re.S)
    if not m:
        raise KeyError(name)
    out = {}
    for i, item in enumerate(x.strip() for x in m.group(1).split(',') if x.strip()):
        ident = item.split('=')[0].strip()
        if '=' in item:
            assert int(item.split('=')[1]) == i, item
        out[ident] = i
    return out


_SIGNATURES = _parse_header(HEADER_PATH)
_SIGNATURES.update(_parse_header(DEBUG_HEADER_PATH))


def header_hash():
    """what csrc/Makefile stamps into the library: sha256 over both headers, first 16 hex digits"""
    import hashlib
    h = hashlib.sha256()
    for p in (HEADER_PATH, DEBUG_HEADER_PATH):
        h.update(open(p, 'rb').read())
    return h.hexdigest()[:16]


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
        if l.ptv_abi_version() != EXPECTED_ABI:
            raise RuntimeError('libptvae_hip.so at %s has ABI version %d, this package expects %d: rebuild it (python -c "import '
                               '__graft_entry__ as g; g.build()")' % (LIB_PATH, l.ptv_abi_version(), EXPECTED_ABI))
        got = l.ptv_header_hash().decode()
        if got != header_hash():
            raise RuntimeError('libptvae_hip.so at %s was built from other headers (hash %s, include/*.h now %s): rebuild it (python -c '
                               '"import __graft_entry__ as g; g.build()")' % (LIB_PATH, got, header_hash()))
        _lib = l
    return _lib


class WgradJob(ctypes.Structure):
    """ptv_wgrad_job of include/ptvae_hip.h (one product of a ptv_wgrad_batch call)"""
    _fields_ = [('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int), ('A', ctypes.c_void_p), ('lda', ctypes.c_long),
                ('B', ctypes.c_void_p), ('ldb', ctypes.c_long), ('C', ctypes.c_void_p), ('ldc', ctypes.c_long), ('alpha', ctypes.c_float),
                ('accumulate', ctypes.c_int), ('dtypes', ctypes.c_int), ('slabs', ctypes.c_int), ('colsum_a', ctypes.c_void_p),
                ('k_top', ctypes.c_void_p), ('k_unit', ctypes.c_long), ('k_rev', ctypes.c_int),
                ('seg_n', ctypes.c_void_p), ('seg_unit', ctypes.c_long), ('seg_period', ctypes.c_int)]


def wgrad_batch(jobs, stream=None):
    """jobs: dicts with the fields of ptv_wgrad_job (tensors for the pointers) -> one ptv_wgrad_batch call on `stream` (default: current)"""
    arr = (WgradJob * len(jobs))()
    for q, j in zip(arr, jobs):
        A, B, C = j['A'], j['B'], j['C']
        q.M, q.N, q.K = j['M'], j['N'], j['K']
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0)
        q.alpha, q.accumulate, q.slabs = j.get('alpha', 1.0), j.get('accumulate', 1), j.get('slabs', 0)
        q.dtypes = (1 if A.dtype == torch.bfloat16 else 0) | (2 if B.dtype == torch.bfloat16 else 0)
        q.colsum_a, q.k_top = ptr(j.get('colsum_a')), ptr(j.get('k_top'))
        q.k_unit, q.k_rev = j.get('k_unit', 0), j.get('k_rev', 0)
        q.seg_n, q.seg_unit, q.seg_period = ptr(j.get('seg_n')), j.get('seg_unit', 0), j.get('seg_period', 0)
    check(lib().ptv_wgrad_batch(arr, len(jobs), stream if stream is not None else stream_ptr()), 'ptv_wgrad_batch')


def prec_code(p):
    return _PREC[p]


_OK_DTYPES = frozenset((torch.float32, torch.bfloat16, torch.int32, torch.int64, torch.uint8, torch.int8))
# Always checked (ptr() runs ~3000 times per train step, so only what costs tens of nanoseconds): the tensor lives on a GPU (no CPU
# fallback), has one of the dtypes the library takes at all, and -- in a process that sees more than one GPU -- lives on the CURRENT
# device (kernels are enqueued on the current device's stream: a tensor of another device would be a wild pointer).
# PTV_PTR_CHECKS=1 (the test suite, tests/conftest.py) checks the device in every process, with messages.
PTR_CHECKS = os.environ.get('PTV_PTR_CHECKS', '0') == '1'
_get_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device
_multi_dev = None


def ptr(t):
    global _multi_dev
    if t is None:
        return None
    if not t.is_cuda or t.dtype not in _OK_DTYPES:
        raise AssertionError('device fp32/bf16/int tensor expected, got %s %s' % (t.device, t.dtype))
    if _multi_dev is None:
        _multi_dev = torch.cuda.device_count() > 1
    if (_multi_dev or PTR_CHECKS) and t.device.index != _get_device():
        raise AssertionError('tensor on %s but the current device is cuda:%d (torch.cuda.set_device first)' % (t.device, _get_device()))
    return t.data_ptr()                       # (a plain int: ctypes converts it for the declared void* parameter)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """raw hipStream_t of the current stream of the current device (torch.cuda.current_stream() builds a Stream object: 2.7 us)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


ERROR_HOOKS = []     # callables run before a failed call raises (functional: a composite that returns early leaves the library's wave-priority marker
                     # wherever it was -- the host's record of it is dropped, so the next product re-sends the state)


def check(rc, what):
    if rc != 0:
        for f in ERROR_HOOKS:
            f()
        raise RuntimeError('libptvae_hip: %s failed with status %d' % (what, rc))


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    check(rc, name)
