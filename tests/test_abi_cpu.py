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
    assert lib.sfk_abi_version() == _lib.ABI_VERSION == 20
    assert lib.sfk_status_string(0) == b"ok" and lib.sfk_status_string(-2).startswith(b"unsupported")


def test_invalid_descriptors_are_rejected_on_the_host(lib):
    from video_classification_amd import _lib
    d = _lib.new_conv_desc()                            # all zero but the handshake: null pointers
    assert lib.sfk_conv_igemm(ctypes.byref(d), None) == -1
    assert lib.sfk_conv_igemm_mtiles(ctypes.byref(d)) == -1
    w = _lib.new_wgrad_desc()
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


def _integration_stub_structs():
    """exec the ctypes snippet of INTEGRATION.md section 2 up to its last struct definition"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", text, flags=re.S).group(1)
    head = code[: code.index("lib.sfk_conv_igemm.argtypes")]
    head = re.sub(r"^lib = .*$", "", head, flags=re.M)
    ns = {"C": ctypes}
    exec(head, ns)
    return ns


def test_integration_md_stub_matches_the_abi(lib):
    """the binding INTEGRATION.md tells a maintainer to write has the library's struct layouts: same field names in the
    same order, same sizes (a struct one pointer short makes the library read past it)"""
    from video_classification_amd import _lib
    ns = _integration_stub_structs()
    for doc_name, ours in (("FMap", _lib._FMap), ("Tap", _lib._Tap), ("BnBwdFuse", _lib._BnBwdFuse),
                           ("ConvEpilogue", _lib._ConvEpilogue), ("ConvDesc", _lib._ConvDesc)):
        doc = ns[doc_name]
        assert [f[0] for f in doc._fields_] == [f[0] for f in ours._fields_], doc_name
        assert ctypes.sizeof(doc) == ctypes.sizeof(ours), doc_name
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert f"sfk_abi_version() == {_lib.ABI_VERSION}" in text


def test_header_struct_sizes_match_ctypes(lib, tmp_path):
    """sizeof() of every struct include/sfk.h declares, from a C compiler, against the ctypes mirrors"""
    import subprocess
    from video_classification_amd import _lib
    names = {"sfk_fmap": _lib._FMap, "sfk_tap": _lib._Tap, "sfk_bn_bwd_fuse": _lib._BnBwdFuse,
             "sfk_conv_epilogue": _lib._ConvEpilogue, "sfk_conv_desc": _lib._ConvDesc, "sfk_wgrad_desc": _lib._WgradDesc, "sfk_stem_src": _lib._StemSrc,
             "sfk_tuning": _lib._Tuning}
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "sfk.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)   # plain C, no HIP
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n, ct in names.items():
        assert int(out[n]) == ctypes.sizeof(ct), n


def test_tuning_table_is_write_once_and_env_free(lib):
    from video_classification_amd import _lib
    d, cur = _lib.new_tuning(), _lib.new_tuning()
    assert lib.sfk_default_tuning(ctypes.byref(d)) == 0
    assert lib.sfk_get_tuning(ctypes.byref(cur)) == 0
    assert (d.igemm_short_k, d.bn_parts, d.nt_reduce_mb, d.nt_bwd_apply_mb, d.pool_blocks) == (5, 1024, 48, 150, 1 << 20)
    assert lib.sfk_init(ctypes.byref(cur)) == 0            # the same table again: fine
    other = _lib._Tuning.from_buffer_copy(cur)
    other.bn_parts = cur.bn_parts + 1
    assert lib.sfk_init(ctypes.byref(other)) == -1         # a different one after load(): rejected, nothing changes
    after = _lib.new_tuning()
    assert lib.sfk_get_tuning(ctypes.byref(after)) == 0
    assert bytes(after) == bytes(cur)
    # the library itself does not read the environment (include/sfk.h "Conventions")
    for f in os.listdir(os.path.join(ROOT, "video-classification_amd", "csrc")):
        assert "getenv" not in open(os.path.join(ROOT, "video-classification_amd", "csrc", f)).read(), f


def _valid_looking_conv_desc(_lib):
    """a descriptor that passes every host-side check (the pointers are never dereferenced by the query)"""
    d = _lib.new_conv_desc()
    for m in (d.x, d.y):
        m.ptr, m.dtype, m.n, m.t, m.h, m.w, m.c, m.ld, m.c_off = 0x1000, 1, 1, 2, 4, 4, 16, 16, 0
    d.rt, d.rh, d.rw = 2, 4, 4
    for a in range(3):
        d.gs[a], d.os[a], d.oo[a] = 1, 1, 0
    d.ntaps, d.w, d.wtaps, d.cin, d.cout = 1, 0x2000, 1, 16, 16
    return d


def test_struct_size_handshake_rejects_other_layouts(lib):
    """SFK_ABI_VERSION stayed 9 across three layout changes in round 2: a binding written for the older sfk_wgrad_desc passed
    the version check and handed the library a struct 48 bytes short.  Since ABI 10 the descriptors carry the caller's
    sizeof: a stale binding gets SFK_ERR_INVALID from every entry point instead of silent garbage."""
    from video_classification_amd import _lib
    d = _valid_looking_conv_desc(_lib)
    assert lib.sfk_conv_igemm_mtiles(ctypes.byref(d)) > 0                 # the same bytes with the right handshake: accepted
    for wrong in (0, ctypes.sizeof(_lib._ConvDesc) - 8, ctypes.sizeof(_lib._ConvDesc) + 48):
        d.struct_size = wrong
        assert lib.sfk_conv_igemm_mtiles(ctypes.byref(d)) == -1
        assert lib.sfk_conv_igemm(ctypes.byref(d), None) == -1
        assert lib.sfk_conv_igemm_family(ctypes.byref(d)) == -1
        assert lib.sfk_conv_bnb_supported(ctypes.byref(d)) == 0
        assert lib.sfk_conv_relu_out_supported(ctypes.byref(d)) == 0
        assert lib.sfk_conv_epilogue_supported(ctypes.byref(d)) == 0
    w = _lib.new_wgrad_desc()
    w.struct_size -= 48                                                     # round 2's pre-dg_w / dg_y layout
    assert lib.sfk_conv_wgrad(ctypes.byref(w), None) == -1
    assert lib.sfk_conv_wgrad_workspace_bytes(ctypes.byref(w)) == -1
    assert lib.sfk_conv_wgrad_dg_supported(ctypes.byref(w)) == 0
    t = _lib.new_tuning()
    t.struct_size += 4
    assert lib.sfk_default_tuning(ctypes.byref(t)) == -1
    assert lib.sfk_get_tuning(ctypes.byref(t)) == -1
    assert lib.sfk_init(ctypes.byref(t)) == -1


def test_header_hash_is_locked_to_the_abi_version():
    """include/sfk.abi = (version, hash of the header's declarations): editing a struct, a prototype or a constant without
    bumping SFK_ABI_VERSION fails here; tools/abi_lock.py --write refuses to re-lock an already locked version number."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("abi_lock", os.path.join(ROOT, "tools", "abi_lock.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    version, digest = mod.header_state()
    assert mod.locked() == (version, digest), \
        "include/sfk.h's declarations changed: bump SFK_ABI_VERSION (and _lib.ABI_VERSION, INTEGRATION.md), then tools/abi_lock.py --write"
    from video_classification_amd import _lib
    assert version == _lib.ABI_VERSION
    # comments and whitespace do not move the hash; a layout edit does
    src = open(mod.HEADER).read()
    assert mod.declarations(src + "\n/* a remark */\n") == mod.declarations(src)
    assert mod.declarations(src.replace("int32_t wtaps, cin, cout;", "int32_t wtaps, cout, cin;", 1)) != mod.declarations(src)


def test_streaming_pointwise_routes_refuse_maps_beyond_32_bit_offsets(lib):
    """the streaming pointwise kernels (conv_pw.hip) read the old rows of a `+=` pass through a 32-bit buffer resource; a map of
    4 GiB or more must stay on the implicit-GEMM kernel (64-bit epilogue pointers) instead of wrapping silently.  Host-side
    routing only: nothing is launched, the pointers are never dereferenced."""
    from video_classification_amd import _lib
    d = _lib.new_conv_desc()
    n, t, h, w, c = 4, 8, 56, 56, 64
    for m in (d.x, d.y):
        m.ptr, m.dtype, m.n, m.t, m.h, m.w, m.c, m.ld, m.c_off = 0x10000, 1, n, t, h, w, c, c, 0
    d.rt, d.rh, d.rw = t, h, w
    for a in range(3):
        d.gs[a], d.os[a], d.oo[a] = 1, 1, 0
    d.ntaps, d.w, d.wtaps, d.cin, d.cout, d.accumulate = 1, 0x20000, 1, c, c, 1
    assert lib.sfk_conv_igemm_family(ctypes.byref(d)) == 3             # K = cout = 64, += : the streaming kernel
    d.y.ld = (1 << 32) // (2 * n * t * h * w) + 8                      # same logical map inside >= 4 GiB of pixel records
    d.y.ld -= d.y.ld % 8
    assert d.y.ld * 2 * n * t * h * w >= (1 << 32) - 64
    assert lib.sfk_conv_igemm_family(ctypes.byref(d)) in (0, 1)        # implicit GEMM, not the 32-bit streaming route


def test_misaligned_vector_operands_are_rejected_on_the_host(lib):
    """sfk_conv_pw_dual reads its filters as packed bf16 pairs, the BatchNorm finalize launches read partial rows as float4:
    an odd pointer must come back as SFK_ERR_INVALID from the host check, not as a misaligned-access fault on the device.
    Nothing is launched (the checks come before the first HIP call), the pointers are never dereferenced."""
    from video_classification_amd import _lib
    maps = []
    for c in (32, 8, 8):
        m = _lib._FMap()
        m.ptr, m.dtype, m.n, m.t, m.h, m.w, m.c, m.ld, m.c_off = 0x10000, 1, 1, 2, 4, 4, c, c, 0
        maps.append(m)
    x1, x2, y = maps
    assert lib.sfk_conv_pw_dual_supported(ctypes.byref(x1), ctypes.byref(x2), ctypes.byref(y)) == 1
    for w1, w2 in ((0x20002, 0x30000), (0x20000, 0x30002), (0x20001, 0x30000)):
        assert lib.sfk_conv_pw_dual(ctypes.byref(x1), w1, ctypes.byref(x2), w2, None, ctypes.byref(y), None) == -1
    PF = ctypes.POINTER(ctypes.c_float)
    f = lambda a: ctypes.cast(a, PF)
    ok = f(0x40000)
    for partials, ws in ((0x50008, None), (0x50000, 0x60004)):
        assert lib.sfk_bn_finalize(f(partials), 4, 8, 100, ok, ok, 1e-5, 0.1, None, None, None, ok, ok, ok, ok,
                                   f(ws) if ws else None, None) == -1
        assert lib.sfk_bn_bwd_finalize(f(partials), 4, 8, 100, ok, ok, ok, ok, ok, f(ws) if ws else None, None) == -1
