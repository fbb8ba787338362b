"""CPU: the C-ABI library builds, loads, and exports every symbol include/ptvae_hip.h declares."""
import ctypes
import os
import re

from polyphonic_chord_texture_disentanglement_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'ptvae_hip.h')).read()
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
    assert l.ptv_abi_version() >= 1
