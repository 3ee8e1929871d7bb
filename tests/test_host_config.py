"""Host-layer configuration object (arseg_amd.ops.config): environment parsing, run-time changes, and the module switches they drive.
CPU only -- nothing here touches the library."""
import pytest


def test_config_from_env(monkeypatch):
    from arseg_amd import ops

    for k in list(__import__("os").environ):
        if k.startswith("ARSEG_"):
            monkeypatch.delenv(k)
    d = ops.Config.from_env()
    assert (d.conv_math, d.conv_autotune, d.conv_find, d.conv_winograd, d.conv_up2_taps, d.conv_range_guard) == ("f16x3", True, "native", True, True, "device")
    assert d.conv_plan_file is None and d.creff_impl == "" and d.creff_tile_rows == 0 and d.lr_subbatch == 0
    monkeypatch.setenv("ARSEG_CONV_MATH", "f32")
    monkeypatch.setenv("ARSEG_CONV_AUTOTUNE", "0")
    monkeypatch.setenv("ARSEG_CONV_RANGE_GUARD", "1")          # the round-2 spelling of the host-synchronising mode
    monkeypatch.setenv("ARSEG_CREFF_TY", "8")
    monkeypatch.setenv("ARSEG_LR_SUBBATCH", "4")
    e = ops.Config.from_env()
    assert (e.conv_math, e.conv_autotune, e.conv_range_guard, e.creff_tile_rows, e.lr_subbatch) == ("f32", False, "host", 8, 4)
    monkeypatch.setenv("ARSEG_CONV_RANGE_GUARD", "0")
    assert ops.Config.from_env().conv_range_guard == "off"


def test_configure_round_trip():
    from arseg_amd import _lib, ops

    before = (ops._WINOGRAD, ops._UP2_TAPS, ops._RANGE_MODE, ops._math)
    prev = ops.configure(conv_winograd=False, conv_up2_taps=False, conv_range_guard="host", conv_math="f32")
    try:
        assert (ops._WINOGRAD, ops._UP2_TAPS, ops._RANGE_MODE, ops._RANGE_GUARD, ops._math) == (False, False, "host", True, _lib.MATH_F32)
        assert ops.set_conv_math("f16x3") == "f32" and ops.config.conv_math == "f16x3"
    finally:
        ops.configure(**prev)
    assert (ops._WINOGRAD, ops._UP2_TAPS, ops._RANGE_MODE, ops._math) == before
    with pytest.raises(_lib.ArsegError):
        ops.configure(no_such_knob=1)


def test_range_word_is_inert_without_a_gpu():
    from arseg_amd import ops

    assert ops.range_tripped(device="cuda:0") is False          # no conv has run: no word exists, nothing is read
