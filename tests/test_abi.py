"""The C-ABI library builds for gfx950, loads, and exports every symbol include/dc_hip.h declares (no GPU calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    ge.build()
    from depth_correction_amd import _native
    return _native


def _declared():
    text = open(os.path.join(ROOT, 'include', 'dc_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dc_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 25
    handle = ctypes.CDLL(built.lib_path())
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    assert handle.dc_version() >= 100


def test_binding_covers_header(built):
    assert set(built._SIGNATURES) == set(_declared())


def test_code_object_is_gfx950(built):
    blob = open(built.lib_path(), 'rb').read()
    targets = set(re.findall(rb'hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_header_is_plain_c():
    src = '#include "dc_hip.h"\nint main(void) { dcSequenceDesc d; (void)d; return DC_Q32 == 2 ? 0 : 1; }\n'
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-x', 'c', '-', '-fsyntax-only'],
                       input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, '_LIB', None)
    monkeypatch.setattr(built, 'lib_path', lambda: '/nonexistent/libdc_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        built.lib()
