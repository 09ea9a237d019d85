"""The C-ABI library loads and exports every symbol include/theora_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "theora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(thip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from theora_amd import _lib
    assert declared_symbols() == sorted(n for n, _, _ in _lib.SYMBOLS)


def test_decoder_api_header_and_exports_agree():
    """include/theoradec_hip.h (the th_decode_* names of theoradec.h:234-322) <-> ctypes
    binding <-> exported symbols."""
    from theora_amd import build, _lib
    text = open(os.path.join(ROOT, "include", "theoradec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(th_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(n for n, _, _ in _lib.DEC_SYMBOLS)
    lib = C.CDLL(build.build())
    for name in declared:
        assert hasattr(lib, name), name


def test_library_exports_every_declared_symbol():
    from theora_amd import build, _lib
    so = build.build()
    lib = C.CDLL(so)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    L = _lib.load()
    assert b"theora_hip" in L.thip_version_string()


def test_argument_validation_without_gpu():
    """Pure host-side argument checks (they return before touching the device)."""
    from theora_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    assert L.thip_state_create(None, 176, 144, 0) == _lib.EFAULT
    for w, hgt, fmt in [(0, 144, 0), (176, 0, 0), (170, 144, 0), (176, 140, 0), (176, 144, 1), (176, 144, 4),
                        (176, 144, -1), (1 << 20, 16, 0)]:
        assert L.thip_state_create(C.byref(h), w, hgt, fmt) == _lib.EINVAL, (w, hgt, fmt)
    assert L.thip_state_create(C.byref(h), 65536 * 8, 65536 * 8, 3) == _lib.EIMPL   # > 2^24 fragments per plane
    assert L.thip_idct8x8_batch(None, None, None, 1) == _lib.EFAULT
    assert L.thip_enc_fdct8x8_batch(1, 1, -1) == _lib.EINVAL
    assert L.thip_enc_frag_metric_batch(99, 1, None, 1, 1, 8, 1, 1, None, 0, 1) == _lib.EINVAL
    assert L.thip_frame_begin(None, 0) == _lib.EFAULT
    assert L.thip_decode_frames(None, None, 0, None, None) == _lib.EFAULT
    assert L.thip_state_ref_idx(None, 0) == _lib.EINVAL
    assert L.thip_state_decode_token_lists(None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_begin(None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_begin_assigned(None, None, None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_finish(None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_open(None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_append(None, 0, 64, None, 0, None, None, None, None) == _lib.EFAULT
    assert L.thip_state_token_lists_abort(None) == _lib.EFAULT
    assert L.thip_state_token_lists_staging(None, None) == _lib.EFAULT


def test_loop_filter_init_slot_matches_oracle_table():
    import numpy as np
    import oracle
    from theora_amd import _lib
    L = _lib.load()
    for fl in range(0, 128):
        bv = np.zeros(256, np.int8)
        L.thip_loop_filter_init(bv.ctypes.data, fl)
        assert np.array_equal(bv, oracle.loop_filter_bv(fl)), fl


def test_product_never_imports_oracle():
    """theora_amd/ and include/ must not reference oracle/ (a product path routed through
    the checker would void every parity claim)."""
    bad = []
    for base in ("theora_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b|oracle/|theora_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


import pytest  # noqa: E402


@pytest.mark.gpu
def test_tile_positions_match_numpy_geometry():
    """thip_state_frag_pos (host table of the library) == the packer's numpy formula
    (thip_state_create allocates device frames, hence the gpu mark)."""
    import theora_amd
    from theora_amd import synth
    for (w, h, fmt) in [(176, 144, 0), (80, 112, 2), (64, 48, 3), (1040, 16, 0)]:
        st = theora_amd.State(w, h, fmt)
        g = synth.Geometry(w, h, fmt)
        assert st.ntiles == g.ntiles and st.tile_off == g.tile_off
        for f in list(range(0, g.nfrags, max(1, g.nfrags // 97))) + [g.nfrags - 1]:
            assert st.frag_pos(f) == g.frag_pos[f]


def test_integration_shim_compiles(tmp_path):
    """The glue INTEGRATION.md asks a libtheora maintainer to write (tests/native/integration_shim.c,
    against a stand-in for the oc_theora_state fields it touches) compiles, warning-free, against
    include/theora_hip.h as it is: the documented drop-in cannot drift away from the ABI."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "integration_shim.c")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-c", src, "-o", str(tmp_path / "shim.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_options_table():
    """thip_set_option / thip_get_option / thip_option_name (include/theora_hip.h): every documented name exists with its
    documented default (when the environment does not say otherwise), values round-trip, unknown names are refused."""
    import ctypes as C
    from theora_amd import _lib
    L = _lib.load()
    names, i = [], 0
    while True:
        hlp = C.c_char_p()
        nm = L.thip_option_name(i, C.byref(hlp))
        if nm is None:
            break
        assert hlp.value
        names.append(nm.decode())
        i += 1
    hdr = open(os.path.join(ROOT, "include", "theora_hip.h")).read()
    for nm in names:
        assert (" " + nm) in hdr, nm          # documented in the header
    for nm in ("fuse", "lanes", "ctx_lanes", "chunk", "skip_static", "lf_sparse", "zerocopy", "wait_spin", "dc_global", "debug",
               "fe_device_dc", "fe_device_tokens", "fe_device_lists", "fe_trace_backend", "fe_prof", "device"):
        assert nm in names, nm
    defaults = dict(fuse=3, lanes=2, ctx_lanes=8, skip_static=1, lf_sparse=-1, zerocopy=1, device=-1, fe_device_lists=-1, fe_device_dc=0)
    for nm, dv in defaults.items():
        if "THIP_" + nm.upper() not in os.environ:
            v = C.c_int(12345)
            assert L.thip_get_option(nm.encode(), C.byref(v)) == 0 and v.value == dv, (nm, v.value)
    v = C.c_int()
    assert L.thip_set_option(b"lf_sparse", 1) == 0 and L.thip_get_option(b"lf_sparse", C.byref(v)) == 0 and v.value == 1
    assert L.thip_set_option(b"lf_sparse", -1) == 0 and L.thip_option(b"lf_sparse") == -1
    assert L.thip_set_option(b"no_such_option", 1) == _lib.EINVAL
    assert L.thip_get_option(b"no_such_option", C.byref(v)) == _lib.EINVAL
    assert L.thip_set_option(None, 1) == _lib.EINVAL


def test_levels_form_host_helpers():
    """The LEVELS form's host-side pieces, no GPU: thip_pack_dequant_table (C) == the numpy packer the tests use; pack_units puts
    every level of a narrow and of a wide block where include/theora_hip.h says (piece j, dword d: x[2j][2d], x[2j][2d+1],
    x[2j+1][2d], x[2j+1][2d+1]; wide: the int16 pieces of the other form over two consecutive units)."""
    import numpy as np
    import theora_amd
    from theora_amd import _lib, synth
    L = _lib.load()
    rng = np.random.default_rng(3)
    tabs = rng.integers(1, 65536, (3, 3, 2, 64)).astype(np.uint16)
    want = theora_amd.pack_dequant_tables(tabs)
    got = np.zeros((18, 64), np.uint16)
    flat = np.ascontiguousarray(tabs).reshape(18, 64)
    for t in range(18):
        L.thip_pack_dequant_table(got[t].ctypes.data, flat[t].ctypes.data)
    assert np.array_equal(got, want)
    # entry (j*8 + c)*2 + p is the factor of natural position (2j + p, c)
    nat = np.zeros(64, np.uint16)
    nat[synth.FZIG_ZAG] = flat[5]
    assert all(got[5][(j * 8 + c) * 2 + p] == nat[(2 * j + p) * 8 + c] for j in range(4) for c in range(8) for p in range(2))
    lv = rng.integers(-127, 128, (70, 64)).astype(np.int16)
    lv[40:] = rng.integers(-32768, 32768, (30, 64))
    wide = np.arange(70) >= 40
    first = np.concatenate([np.arange(40), 40 + 2 * np.arange(30)])
    buf = theora_amd.pack_units(lv, wide, first, 100)
    assert buf.size == 2 * 4096

    def at(unit, piece):
        o = (unit >> 6) * 4096 + piece * 1024 + (unit & 63) * 16
        return buf[o:o + 16]
    for b in (0, 17, 39):
        x = lv[b].reshape(8, 8)
        for j in range(4):
            p = at(int(first[b]), j).view(np.int8)
            for d in range(4):
                assert list(p[4 * d:4 * d + 4]) == [x[2 * j, 2 * d], x[2 * j, 2 * d + 1], x[2 * j + 1, 2 * d], x[2 * j + 1, 2 * d + 1]]
    for b in (40, 55, 69):
        x = lv[b].reshape(8, 8)
        for q in range(8):
            j, h = q >> 1, q & 1
            p = at(int(first[b]) + (q >> 2), q & 3).view(np.int16)
            assert list(p) == [x[2 * j + k, 4 * h + cc] for cc in range(4) for k in range(2)]


def test_no_kernel_owns_scratch():
    """k_recon_lf<false> picked up 16 bytes of scratch a lane unnoticed.  Every build records the compiler's
    resource remarks (theora_amd/build.py -> libtheora_hip.so.resources.json); no kernel of the product may own scratch, and the
    reconstruction kernels keep the occupancy their LDS budget is sized for."""
    from theora_amd import build
    rows = build.resources()
    by = {r["name"]: r for r in rows if r.get("name")}
    assert len(by) >= 55, "the resource table of the build is incomplete: %d kernels" % len(by)
    bad = {n: r["ScratchSize"] for n, r in by.items() if r.get("ScratchSize", 0) != 0}
    assert not bad, "kernels with scratch: %r" % bad
    for n in ("k_recon_lf<true>", "k_recon_lf<false>"):
        assert by[n]["VGPRs"] <= 96 and by[n]["Occupancy"] >= 5 and by[n]["LDS"] <= 7168, (n, by[n])
    for n in ("k_recon_lf_sb<true>", "k_recon_lf_sb<false>"):
        assert by[n]["Occupancy"] >= 8, (n, by[n])
