"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every function include/sfk.h declares
(no compute calls here); host-side argument validation is reachable without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from video_classification_amd import _lib
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sfk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sfk_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from video_classification_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sfk.h but not exported by libsfk.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_struct_layout_matches_header(lib):
    from video_classification_amd import _lib
    assert ctypes.sizeof(_lib._FMap) == 40            # void* + 8 x int32
    assert ctypes.sizeof(_lib._Tap) == 4
    assert _lib._ConvDesc.taps.size == 4 * _lib.SFK_MAX_TAPS
    assert lib.sfk_abi_version() == 3
    assert lib.sfk_status_string(0) == b"ok" and lib.sfk_status_string(-2).startswith(b"unsupported")


def test_invalid_descriptors_are_rejected_on_the_host(lib):
    from video_classification_amd import _lib
    d = _lib._ConvDesc()                                # all zero: null pointers
    assert lib.sfk_conv_igemm(ctypes.byref(d), None) == -1
    assert lib.sfk_conv_igemm_mtiles(ctypes.byref(d)) == -1
    w = _lib._WgradDesc()
    assert lib.sfk_conv_wgrad(ctypes.byref(w), None) == -1
    assert lib.sfk_fc_fwd(None, None, None, None, 1, 1, 1, None) == -1
    assert lib.sfk_adam(None, None, None, None, 10, 0.1, 0.9, 0.999, 1e-8, 1.0, None, None, 0, None) == -1


def test_missing_library_fails_loudly(tmp_path):
    from video_classification_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.SfkError):
            _lib.load(str(tmp_path / "libsfk.so"))
    finally:
        _lib._lib = saved
