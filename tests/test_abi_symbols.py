"""CPU: the C-ABI library builds, loads, and exports every symbol include/ptvae_hip.h declares."""
import ctypes
import os
import re

from polyphonic_chord_texture_disentanglement_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = ''.join(open(os.path.join(ROOT, 'include', h)).read() for h in ('ptvae_hip.h', 'ptvae_hip_debug.h'))
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ptv_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    names = _declared()
    assert len(names) >= 5
    l = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(l, n), 'missing export %s' % n
    # the python binding table covers the header exactly
    assert sorted(_lib.exported_symbols()) == names


def test_identification_calls_work_without_gpu():
    l = _lib.lib()
    assert l.ptv_arch() == b'gfx950'
    from polyphonic_chord_texture_disentanglement_amd._lib import EXPECTED_ABI
    assert l.ptv_abi_version() == EXPECTED_ABI
    assert l.ptv_header_hash().decode() == _lib.header_hash()                  # the library was built from the headers in the tree


def test_product_header_carries_no_debug_hooks():
    src = open(os.path.join(ROOT, 'include', 'ptvae_hip.h')).read()
    assert not re.search(r'\bptv_(debug|prof)_[a-z0-9_]+\s*\(', src)


def integration_snippet():
    """the worked ctypes example of INTEGRATION.md section 2 (the code block that binds ptv_gemm), verbatim"""
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, flags=re.S)
    code = [b for b in blocks if 'lib.ptv_gemm.argtypes' in b]
    assert len(code) == 1
    return code[0]


def test_integration_md_example_matches_the_header():
    """the documented binding passed 18 arguments to a 19-argument entry point.  Build the argtypes the snippet
    declares (no GPU needed: the .so loads on CPU) and compare them with the table generated from include/ptvae_hip.h."""
    code = integration_snippet()
    ns = {}
    head = code.split('def linear')[0].replace('polyphonic_chord_texture_disentanglement_amd/csrc/libptvae_hip.so', _lib.LIB_PATH)
    exec(head, ns)                                            # imports, CDLL, restype / argtypes
    res, args = _lib._SIGNATURES['ptv_gemm']
    assert list(ns['lib'].ptv_gemm.argtypes) == list(args) and ns['lib'].ptv_gemm.restype is res
    call = re.search(r'lib\.ptv_gemm\((.*?)\)\n\s+assert rc == 0', code, flags=re.S).group(1)
    call = re.sub(r'#.*', '', call)
    depth, n = 0, 1
    for ch in call:
        depth += ch in '([' 
        depth -= ch in ')]'
        n += (ch == ',' and depth == 0)
    assert n == len(args) == 19
