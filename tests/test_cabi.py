"""CPU: the C-ABI shared library loads and exports exactly what include/tf_msda.h declares;
argument validation works without a GPU (no compute is launched here)."""
import ctypes
import os
import re

import pytest

from trackformer_amd import _cabi, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_all()
    return _cabi.lib()


def _declared_symbols():
    names = set()
    for header in sorted(os.listdir(os.path.join(REPO, "include"))):
        if not header.endswith(".h"):
            continue
        text = open(os.path.join(REPO, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_cabi.EXPORTED_SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(_cabi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), name


def test_abi_version_and_strerror(lib):
    assert lib.tf_msda_abi_version() == _cabi.ABI_VERSION
    assert lib.tf_msda_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert len(lib.tf_msda_strerror(code)) > 0


def test_null_and_bad_dims_are_rejected_before_any_gpu_work(lib):
    shp = (ctypes.c_int64 * 2)(2, 2)
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.tf_msda_forward_f32(None, shp, one, one, one, 1, 4, 1, 4, 1, 1, 1, None) == -1
    assert lib.tf_msda_forward_f32(one, None, one, one, one, 1, 4, 1, 4, 1, 1, 1, None) == -1
    assert lib.tf_msda_forward_f32(one, shp, one, one, one, 0, 4, 1, 4, 1, 1, 1, None) == -2
    assert lib.tf_msda_forward_f32(one, shp, one, one, one, 1, 4, 1, 4, 17, 1, 1, None) == -2
    assert lib.tf_msda_forward_f32(one, shp, one, one, one, 1, 5, 1, 4, 1, 1, 1, None) == -3
    assert lib.tf_msda_backward_f64(one, shp, one, one, one, one, one, None, 1, 4, 1, 4, 1, 1, 1,
                                    None) == -1
    assert lib.tf_msda_backward_f32(one, shp, one, one, one, one, one, one, 1, 5, 1, 4, 1, 1, 1,
                                    None) == -3
    with pytest.raises(_cabi.MSDAError):
        _cabi.check(-3, "unit test")


def test_option_knobs_round_trip_without_a_gpu(lib):
    """tf_msda_set_option / tf_msda_set_tiled only touch host state: names, previous values, unknown names."""
    int_min = -2 ** 31
    assert lib.tf_msda_set_option(b"no_such_option", 1) == int_min
    assert lib.tf_msda_set_option(None, 1) == int_min
    prev = lib.tf_msda_set_option(b"quad_lds_kb", 48)
    assert prev == 40                                     # the shipped default (4 workgroups per CU)
    assert lib.tf_msda_set_option(b"quad_lds_kb", prev) == 48
    for name, default in ((b"quad_ta_mask", 0), (b"quad_waves", 4), (b"quad_npass", 3), (b"quad_split", 1)):
        assert lib.tf_msda_set_option(name, default) == default
    before = lib.tf_msda_set_tiled(0)
    assert before == -1                                   # -1: follow TF_MSDA_TILED (unset: the quad kernel)
    assert lib.tf_msda_set_option(b"tiled", 2) == 0
    assert lib.tf_msda_set_tiled(-1) == 2


def test_pquad_and_linear_knobs_round_trip_without_a_gpu(lib):
    """The persistent encoder kernel's options and the split GEMM's block-shape variant are host state as well."""
    int_min = -2 ** 31
    for name, default in ((b"pquad", 1), (b"pquad_wide", 1), (b"pquad_npass", 2), (b"pquad_lds_kb", 52),
                          (b"pquad_halo_y", 6), (b"pquad_halo_x", 10), (b"pquad_wg_per_cu", 3),
                          (b"pquad_prefetch", 0), (b"pquad_skew", 0)):
        assert lib.tf_msda_set_option(name, default) == default, name
    assert lib.tf_msda_set_option(b"pquad_no_such", 1) == int_min
    prev = lib.tf_msda_set_option(b"linear_stream_ti", 3)
    assert prev in (0, 3)                                 # 0: per-shape choice (the default)
    assert lib.tf_msda_set_option(b"linear_stream_ti", prev) == 3
    # knobs of experiments that were measured and removed are unknown names now (include/tf_msda.h lists what is left)
    for gone in (b"linear_variant", b"linear_bufstore", b"linear_deep", b"conv3_bufload", b"linear_astat"):
        assert lib.tf_msda_set_option(gone, 1) == int_min, gone
