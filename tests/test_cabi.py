"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    import glob
    syms = set()
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms |= set(re.findall(r"\b(irsde_[a-z0-9_]+)\s*\(", src))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 17
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "missing export %s" % s
    assert sorted(_lib.SYMBOLS) == syms  # the ctypes binding covers the whole header


def test_version_and_error_string():
    L = _lib.lib()
    assert L.irsde_version() == 107
    assert isinstance(L.irsde_last_error(), bytes)


def _create(nf=64, depth=4, in_nc=3, out_nc=3):
    L = _lib.lib()
    cfg = _lib.Config(in_nc, out_nc, nf, depth, 0, 0)
    h = ctypes.c_void_p()
    rc = L.irsde_create(ctypes.byref(cfg), ctypes.byref(h))
    return L, rc, h


def test_graft_entry_build_passes():
    """The driver's "does it build" check: __graft_entry__.build() compiles (or finds) the library and checks that its ABI version is the
    last entry of the header's changelog (a stale literal there failed the check once the ABI moved to 103)."""
    import __graft_entry__ as g
    path = g.build()
    assert os.path.exists(path) and path.endswith("libirsde_hip.so")


def test_engine_inventory_matches_reference_state_dict():
    """irsde_create is host-only: the weight inventory must be the reference's 151 state_dict tensors."""
    L, rc, h = _create()
    assert rc == 0
    try:
        n = L.irsde_num_weights(h)
        shapes = O.unet_param_shapes(3, 3, 64, 4)
        assert n == len(shapes) == 151
        for i in range(n):
            name = L.irsde_weight_name(h, i).decode()
            shp = (ctypes.c_int64 * 4)()
            nd = ctypes.c_int()
            assert L.irsde_weight_shape(h, i, shp, ctypes.byref(nd)) == 0
            assert tuple(shp[: nd.value]) == tuple(shapes[name]), name
    finally:
        L.irsde_destroy(h)


@pytest.mark.parametrize("kw", [dict(nf=48), dict(depth=0), dict(in_nc=5), dict(nf=64, depth=6)])
def test_create_rejects_bad_config(kw):
    L, rc, h = _create(**kw)
    assert rc != 0
    assert len(L.irsde_last_error()) > 0


def test_load_weight_errors():
    L, rc, h = _create(nf=32, depth=2)
    assert rc == 0
    try:
        buf = (ctypes.c_float * 16)()
        shp = (ctypes.c_int64 * 1)(16)
        assert L.irsde_load_weight(h, b"no.such.weight", buf, shp, 1) == -4
        assert b"unknown weight" in L.irsde_last_error()
        assert L.irsde_load_weight(h, b"final_conv.bias", buf, shp, 1) == -4  # wrong shape (3,)
        assert b"shape mismatch" in L.irsde_last_error()
        # state errors: sampling before weights/schedule
        assert L.irsde_sample(h, 0, buf, buf, None, 0, 0, 1, 8, 8, 1, 0, buf, None, 0) != 0
        assert L.irsde_set_schedule(h, 10, buf) != 0
    finally:
        L.irsde_destroy(h)


def test_plain_c_host(tmp_path):
    """include/irsde_hip.h is valid C99 and a plain C program (no HIP / torch headers) can drive the library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "cabi_host")
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c", "cabi_host.c"), "-o", exe, "-L", libdir, "-lirsde_hip",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c host ok" in r.stdout
