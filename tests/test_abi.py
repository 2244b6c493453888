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


def _prototypes():
    """{name: (return type, [parameter declarations])} of every function include/dc_hip.h declares."""
    text = open(os.path.join(ROOT, 'include', 'dc_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'\b([A-Za-z_][A-Za-z0-9_]*(?:\s*\*)?)\s+(dc_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', text, flags=re.S):
        params = [p.strip() for p in m.group(3).replace('\n', ' ').split(',')]
        out[m.group(2)] = (m.group(1).strip(), [] if params in (['void'], ['']) else params)
    return out


def _ctype_of(decl):
    """ctypes type a C parameter declaration must be bound with."""
    if '*' in decl or 'dcStream_t' in decl:
        return ctypes.c_void_p
    if decl.startswith('int64_t'):
        return ctypes.c_int64
    if decl.startswith('size_t'):
        return ctypes.c_size_t
    if decl.startswith('double'):
        return ctypes.c_double
    if decl.startswith('int ') or decl.startswith('int32_t'):
        return ctypes.c_int
    raise AssertionError('unhandled parameter type: %r' % decl)


def test_binding_signatures_match_header(built):
    """Every ctypes signature has the header's parameter list: same length, same kind per position, same return."""
    protos = _prototypes()
    assert set(protos) == set(built._SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = built._SIGNATURES[name]
        assert len(args) == len(params), (name, len(args), len(params))
        for pos, (a, decl) in enumerate(zip(args, params)):
            assert a is _ctype_of(decl), (name, pos, decl, a)
        assert res is _ctype_of(ret + ' '), (name, ret, res)


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


def test_ab_switches_are_locked_unless_the_process_asks_for_them():
    """dc_set_option / dc_knn_set_shell_budget / dc_knn_set_fine_cell_count / dc_features_set_tiled (A-B measurements, path-against-path tests) are refused
    (DC_ERR_UNSUPPORTED) in a process without DC_ENABLE_ABLATIONS=1: the product library has no mutable process-wide state.  Host
    functions: no GPU needed."""
    import subprocess
    import sys
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); "
            "print(lib.dc_set_option(3, 0), lib.dc_knn_set_shell_budget(1000), lib.dc_knn_set_fine_cell_count(14)); "
            "lib.dc_features_set_tiled(0); print(lib.dc_features_set_tiled(1))")       # (the second call reports what the first left)
    from depth_correction_amd import _native as nv
    path = nv.lib_path()
    for flag, want in ((None, '-4 -4 -4\n1'), ('0', '-4 -4 -4\n1'), ('1', '0 0 0\n0')):
        env = {k: v for k, v in os.environ.items() if k != 'DC_ENABLE_ABLATIONS'}
        if flag is not None:
            env['DC_ENABLE_ABLATIONS'] = flag
        out = subprocess.run([sys.executable, '-c', code, path], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want, (flag, out.stdout, out.stderr)
