"""Host-side invariants of the convolution launch geometry (csrc/conv_gemm.hip gemm_geometry, csrc/conv_skinny.hip
conv_skinny_geometry) against the GroupNorm statistics layout the plan reserves and the consumers read
(csrc/spdm_api.hip Ctx::salloc, kernels.h StatsRef).  No GPU: spdm_debug_geometry is pure host arithmetic.

A launch that writes its statistics in one slot layout while the plan tells the consumer another is a silent wrong-result
bug that only some batch sizes reach (round 2: the width-2 rule with ksplit == 1); these checks run over every layer
shape of the U-Net at every batch size class, on the CPU."""
import ctypes

import pytest

from state_policy_diffusionmodel_amd import _lib

SPLITK_WORKSPACE_BYTES = 48 << 20
FIELDS = ("m_tile", "n_tile", "n_tiles", "slots", "ksplit", "skinny", "st_m_tile", "st_n_tiles", "reserved", "combine_rows")


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def geometry(lib, M, N, K, HW, W, taps, sw=0):
    out = (ctypes.c_int32 * 10)()
    rc = lib.spdm_debug_geometry(M, N, K, HW, W, taps, sw, ctypes.byref(out))
    assert rc == 0, rc
    g = dict(zip(FIELDS, out))
    g["reg"] = g["skinny"] >> 1           # out[5]: bit 0 = conv_skinny.hip, bit 1 = conv_reg.hip
    g["skinny"] &= 1
    return g


def stats_slots(HW, m_tile, n_tiles):
    return ((HW + m_tile - 2) // m_tile + 1) * n_tiles


def unet_conv_shapes(H, D):
    """(HW, W, N, K, taps) of every 3x3 conv of models/unet.py's UNet_Film on an (H, D) state map (padded to multiples of 8
    like the product does): inc, down1-3, up1-3 DoubleConvs (models/unet.py:186-226)."""
    Hp, Wp = (H + 7) // 8 * 8, (D + 7) // 8 * 8
    lv = [(Hp * Wp // 4 ** i, Wp >> i) for i in range(4)]
    taps = [9 if w > 1 else 3 for _, w in lv]
    shapes = []
    def dc(level, cin, cmid, cout):
        hw, w = lv[level]
        shapes.append((hw, w, cmid, cin, taps[level]))
        shapes.append((hw, w, cout, cmid, taps[level]))
    dc(0, 64, 64, 64)                                      # inc (second conv onwards; the first is conv_in_kernel)
    for level, cin, cout in ((1, 64, 128), (2, 128, 256), (3, 256, 256)):       # Down: DoubleConv(residual) + DoubleConv
        dc(level, cin, cin, cin)
        dc(level, cin, cout, cout)
    for level, cin, cout in ((2, 512, 128), (1, 256, 64), (0, 128, 64)):        # Up: DoubleConv(residual) + DoubleConv(mid = cin / 2)
        dc(level, cin, cin, cin)
        dc(level, cin, cin // 2, cout)
    return [s for s in shapes if s[3] % 32 == 0]


BATCHES = [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 48, 64, 96, 100, 128, 192, 255, 256, 257, 384, 400, 450, 510, 512, 640, 768, 1000, 1024, 1536,
           2048, 3000, 4096, 8192]


@pytest.mark.parametrize("H,D", [(16, 5), (32, 5), (64, 6), (8, 5), (24, 5), (64, 12)])
def test_statistics_layout_matches_the_launch_for_every_layer_and_batch(lib, H, D):
    seen = set()
    for HW, W, N, K, taps in unet_conv_shapes(H, D):
        for B in BATCHES:
            M = B * HW
            g = geometry(lib, M, N, K, HW, W, taps)
            tag = (HW, W, N, K, taps, B, tuple(g.items()))
            # the tile grid covers N exactly, with tile shapes the kernels instantiate
            assert g["n_tile"] * g["n_tiles"] == N, tag
            assert g["n_tile"] in (16, 32, 64, 128), tag
            assert g["m_tile"] in (16, 32, 64, 128, 256, 512), tag
            assert g["ksplit"] >= 1, tag
            # conv_reg.hip: 64 -> 64 channels on width-8 / width-4 maps, from one wave tile per CU up; one slot per 64-row wave tile
            assert bool(g["reg"]) == (N == 64 and K == 64 and W in (4, 8) and taps == 9 and HW % 64 == 0 and M // 64 >= 256), tag
            if g["reg"]:
                assert (g["m_tile"], g["n_tile"], g["ksplit"], g["skinny"]) == (64, 64, 1, 0), tag
            # the statistics layout is the launch's own tiling, or the combine kernel's when (and only when) there is a combine pass
            if g["ksplit"] > 1:
                assert not g["skinny"], tag
                assert (g["st_m_tile"], g["st_n_tiles"]) == (g["combine_rows"], 1), tag
                assert K % 32 == 0 and g["ksplit"] <= K // 32, tag
                # tilings conv_wide.hip takes (128-wide, rows per sample % 4 == 0, K % 64 == 0) walk their chunks in pairs:
                # a finer split would hand some workgroups an empty chunk range (down1.dc2a at B = 383..510)
                if g["n_tile"] == 128 and HW % 4 == 0 and K % 64 == 0:
                    assert g["ksplit"] <= K // 64, tag
                assert g["ksplit"] * M * N * 4 <= SPLITK_WORKSPACE_BYTES, tag
                assert HW % g["combine_rows"] == 0 and (g["combine_rows"] * N <= 2048 or g["combine_rows"] % 2 == 1), tag
            else:
                assert (g["st_m_tile"], g["st_n_tiles"]) == (g["m_tile"], g["n_tiles"]), tag
            assert g["slots"] == stats_slots(HW, g["st_m_tile"], g["st_n_tiles"]), tag
            # ... and fits what the plan reserved for this (HW, C) at ANY batch size
            assert g["slots"] <= g["reserved"], tag
            if g["skinny"]:
                assert g["m_tile"] <= 64 and g["n_tile"] in (16, 32, 64), tag
            seen.add((g["m_tile"], g["n_tile"], g["ksplit"] > 1, bool(g["skinny"])))
    # the sweep is only meaningful if it reaches the regimes the product has: skinny, split-K, plain 128- and 256-row tiles
    if (H, D) == (64, 6):
        assert any(s[3] for s in seen) and any(s[2] for s in seen)
        assert any(s[0] == 256 and not s[2] for s in seen) and any(s[0] == 128 and not s[2] and not s[3] for s in seen)


def test_reservation_does_not_depend_on_the_batch(lib):
    """Ctx::salloc sizes the statistics buffer once per handle (max_batch), the geometry is chosen per call: the reserved slot
    count must be a function of (HW, N) alone."""
    for HW, W, N, K, taps in unet_conv_shapes(64, 6):
        reserved = {geometry(lib, B * HW, N, K, HW, W, taps)["reserved"] for B in BATCHES}
        # the only batch-dependent term is the launch's own slot count, which the fixed terms must already cover
        fixed = max(stats_slots(HW, 128, max(1, N // 64)), stats_slots(HW, 16, max(1, N // 16)),
                    stats_slots(HW, geometry(lib, HW, N, K, HW, W, taps)["combine_rows"], 1), stats_slots(HW, max(HW // 4, 1), 1))
        assert reserved == {fixed}, (HW, N, K, reserved, fixed)


def test_switches_change_the_geometry_but_keep_the_invariants(lib):
    NO_SPLITK, NO_SKINNY, NO_WIDE, NO_REG64 = 1 << 14, 1 << 17, 1 << 0, 1 << 28
    for sw in (NO_SPLITK, NO_SKINNY, NO_SPLITK | NO_SKINNY, NO_WIDE, NO_REG64):
        for HW, W, N, K, taps in unet_conv_shapes(64, 6):
            for B in (1, 8, 64, 256, 512, 1024, 4096):
                g = geometry(lib, B * HW, N, K, HW, W, taps, sw)
                if sw & NO_SPLITK:
                    assert g["ksplit"] == 1
                if sw & NO_SKINNY:
                    assert not g["skinny"]
                if sw & NO_REG64:
                    assert not g["reg"]
                if g["ksplit"] > 1:
                    assert (g["st_m_tile"], g["st_n_tiles"]) == (g["combine_rows"], 1)
                else:
                    assert (g["st_m_tile"], g["st_n_tiles"]) == (g["m_tile"], g["n_tiles"])
                assert g["slots"] == stats_slots(HW, g["st_m_tile"], g["st_n_tiles"]) <= g["reserved"]


def test_bad_arguments_are_refused(lib):
    out = (ctypes.c_int32 * 10)()
    assert lib.spdm_debug_geometry(100, 64, 64, 48, 8, 9, 0, ctypes.byref(out)) != 0      # M not a multiple of HW
    assert lib.spdm_debug_geometry(0, 64, 64, 48, 8, 9, 0, ctypes.byref(out)) != 0
