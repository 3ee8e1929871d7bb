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


def test_gemm_x3_knob(monkeypatch):
    """ARSEG_CONV_GEMM_X3 = 1 | wino | 0 and what each enables: the split-row chain needs the knob fully on, the f16x3 back end and the device
    (or no) range guard -- under the host-synchronising guard or another back end the producers keep writing fp32 tensors."""
    from arseg_amd import ops

    for v, want in (("1", True), ("wino", "wino"), ("0", False)):
        monkeypatch.setenv("ARSEG_CONV_GEMM_X3", v)
        assert ops.Config.from_env().conv_gemm_x3 == want
    monkeypatch.delenv("ARSEG_CONV_GEMM_X3")
    assert ops.Config.from_env().conv_gemm_x3 is True
    prev = ops.configure(conv_gemm_x3=True, conv_math="f16x3", conv_range_guard="device")
    try:
        assert ops.gemm_x3_enabled()
        for kw in ({"conv_gemm_x3": "wino"}, {"conv_gemm_x3": False}, {"conv_math": "f32"}, {"conv_range_guard": "host"}):
            old = ops.configure(**kw)
            assert not ops.gemm_x3_enabled(), kw
            ops.configure(**old)
        assert ops.gemm_x3_enabled()
    finally:
        ops.configure(**prev)


def test_split_rows_layout_matches_host_packer():
    """ops.SplitRows (the activation operand format of arseg_gemm_x3_fwd) is the f16x3 weight format of arseg_split_weight_f16x3_host: per 32
    values 32 hi halves then 32 lo halves, hi + lo = x to 22 bits.  Checked on the CPU with the host packer and SplitRows.float()."""
    import ctypes

    import numpy as np
    import torch

    from arseg_amd import _lib, ops

    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(7))
    rows, K = 24, 96
    x = (g.standard_normal((rows, K)) * np.exp(g.standard_normal((rows, 1)))).astype(np.float32)
    out = np.empty_like(x)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    _lib.check(lib.arseg_split_weight_f16x3_host(P(x), rows, K, P(out), None), "split_weight_f16x3")
    halves = out.view(np.float16).reshape(rows, K // 32, 2, 32).astype(np.float32)
    hi, lo = halves[:, :, 0].reshape(rows, K), halves[:, :, 1].reshape(rows, K)
    assert np.all(np.abs(hi) <= np.abs(x)) and np.all(np.abs(hi + lo - x) <= 2.0 ** -20 * np.abs(x) + 2.0 ** -24)      # (the host packer truncates both halves)
    sr = ops.SplitRows(torch.from_numpy(out).reshape(2, 3, 4, K))
    assert sr.shape == (2, 3, 4, K)
    assert torch.equal(sr.float(), torch.from_numpy(hi + lo).reshape(2, 3, 4, K))


def test_config_validation_is_the_same_for_env_and_configure(monkeypatch):
    """ A typo in an A/B run must not silently measure the default kernel: the environment is held to configure()'s rules."""
    from arseg_amd import _lib, ops

    for var, bad in (("ARSEG_CREFF_WARP_IMPL", "rol"), ("ARSEG_CONV_WINO_MARGIN", "0"), ("ARSEG_CONV_WINO_MARGIN", "-1"), ("ARSEG_CONV_MATH", "fp32"),
                     ("ARSEG_CREFF_TY", "12"), ("ARSEG_CONV_RANGE_GUARD", "devcie"), ("ARSEG_CREFF_MAX_WGS", "-3"), ("ARSEG_LR_SUBBATCH", "x")):
        monkeypatch.setenv(var, bad)
        with pytest.raises(_lib.ArsegError):
            ops.Config.from_env()
        monkeypatch.delenv(var)
    for kw in ({"creff_warp_impl": "rol"}, {"conv_wino_margin": 0}, {"conv_math": "fp32"}, {"creff_tile_rows": 12}, {"creff_max_wgs": -1}, {"conv_range_guard": "x"}):
        with pytest.raises(_lib.ArsegError):
            ops.configure(**kw)
    assert ops.Config.from_env() == ops.Config()


def test_creff_dispatch_table():
    """ONE dispatch story for the fused warp + CReFF entry point, stated by the library itself (arseg_creff_warp_select, a
    pure query: no GPU needed) -- DESIGN.md 5.2 holds the same table:
        rolling kernel  (creff_roll.hip)  C = 64, 7 x 7, no head or <= 16 classes, schedule fits its piece table
        tile kernel     (creff_rr.hip)    17-32 classes, schedules the rolling kernel does not admit, impl = TILES
        two kernels     (warp_mvq + creff_mfma / creff)   everything else (C != 64, other windows)
    and the advisor's example of round 4 (fixed 128-row segments of a 720 x 960 x 11 launch on 8 workgroups = 83 pieces per workgroup, more than the
    64-entry table) is refused by the rolling kernel instead of being computed in part."""
    from arseg_amd import _lib

    sel = _lib.load().arseg_creff_warp_select
    AUTO, TILES, ROLL = 0, 1, 2
    q = lambda N, C, Hp, Wp, n_cls, impl=AUTO, seg=0, wgs=0, k=7: sel(N, C, Hp, Wp, Hp // 2, Wp // 2, k, k, n_cls, impl, seg, wgs)      # noqa: E731
    assert q(11, 64, 512, 1024, 12) == ROLL and q(11, 64, 512, 1024, 0) == ROLL and q(11, 64, 512, 1024, 16) == ROLL
    assert q(3, 64, 1024, 2048, 12) == ROLL and q(1, 64, 7, 9, 0) == ROLL
    assert q(11, 64, 512, 1024, 19) == TILES and q(11, 64, 512, 1024, 32) == TILES                  # 17-32 classes: the tile kernel
    assert q(11, 64, 512, 1024, 19, ROLL) == _lib.ARSEG_EUNSUPPORTED                                # ... and never the rolling kernel
    assert q(11, 64, 512, 1024, 12, TILES) == TILES
    assert q(11, 64, 720, 960, 12, AUTO, 128, 8) == TILES                                           # 83 pieces per workgroup
    assert q(11, 64, 720, 960, 12, ROLL, 128, 8) == _lib.ARSEG_EUNSUPPORTED
    assert q(11, 64, 720, 960, 12, AUTO, 128, 0) == ROLL
    for args in ((11, 256, 128, 256, 19), (11, 64, 512, 1024, 33), (33, 64, 64, 64, 12)):             # not the fused entry point's shapes
        assert q(*args) == _lib.ARSEG_EUNSUPPORTED
    assert q(11, 64, 512, 1024, 12, k=5) == _lib.ARSEG_EUNSUPPORTED
    assert q(11, 64, 512, 1024, 12, 7) == _lib.ARSEG_EINVAL and q(0, 64, 8, 8, 0) == _lib.ARSEG_EINVAL
