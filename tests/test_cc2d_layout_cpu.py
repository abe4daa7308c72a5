"""CPU: the index arithmetic the 2-D pair pass (xcorr_amd/csrc/cc2d.hip) relies on, restated in numpy and checked
exhaustively for small sizes -- no GPU, no library call.

* `t2_off`: the row-group layout of the spectra S2 / the intermediate T2 is a bijection of the [P1][pitch] block, and it
  is additive over the butterfly strides (the column kernel adds a per-thread offset to a per-butterfly-input base).
* the row kernel's load addressing: one per-lane offset plus a uniform stride reproduces `t2_off`.
* the arg-max's validity rule: output u = 2m + h of the padded C2R row lands in column (u + L - 1) mod P2 of the
  fftshift-ed map, which is one of the 2L - 1 lags exactly when NOT (L <= u <= P2 - L) -- one unsigned compare on u - L.
"""
import numpy as np
import pytest


def t2_off(m1, k2, pitch, rb):
    m1 = np.asarray(m1, np.int64); k2 = np.asarray(k2, np.int64)
    if rb == 1:
        return m1 * pitch + k2
    return ((m1 // rb) * (pitch >> 4) + (k2 >> 4)) * (16 * rb) + (m1 % rb) * 16 + (k2 & 15)


@pytest.mark.parametrize('P1', [16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize('RB', [1, 4, 32, 512])
def test_row_group_layout_is_a_bijection_of_the_block(P1, RB):
    rb = min(RB, P1)
    for R in (32, 100 // 4 * 4 + 28, 4096 if P1 == 512 else 256):
        pitch = (R + 1 + 15) // 16 * 16
        m1, k2 = np.meshgrid(np.arange(P1), np.arange(pitch), indexing='ij')
        off = t2_off(m1, k2, pitch, rb).ravel()
        assert off.min() == 0 and off.max() == P1 * pitch - 1
        assert np.unique(off).size == off.size


@pytest.mark.parametrize('P1,stride', [(128, 16), (128, 32), (256, 32), (256, 64), (512, 64)])
def test_layout_is_additive_over_the_butterfly_strides(P1, stride):
    """t2_off(j + t S, k2) = t2_off(j, k2) + t2_off(t S, 0) for j < S (S and the group size are powers of two)."""
    pitch = 4112
    for rb in (1, 8, 32, P1):
        rb = min(rb, P1)
        j = np.arange(stride)[:, None, None]; t = np.arange(P1 // stride)[None, :, None]; k2 = np.array([0, 5, 16, 4096])[None, None, :]
        lhs = t2_off(j + t * stride, k2, pitch, rb)
        rhs = t2_off(j, k2, pitch, rb) + t2_off(t * stride, 0, pitch, rb)
        assert np.array_equal(lhs, rhs)


@pytest.mark.parametrize('P1', [16, 64, 512])
def test_row_kernel_addressing(P1):
    """k2d_rows_inv: element (m1, k) at row_base + ((k >> 4) << (4 + rbl)) + (k & 15); the loads of thread tid are
    lane_off + it * it_stride with k = tid + it * NT."""
    NT, R = 256, 4096
    pitch = (R + 1 + 15) // 16 * 16
    for RB in (1, 8, 512):
        rb = min(RB, P1); rbl = int(np.log2(rb))
        for m1 in (0, 1, rb - 1, P1 // 2 + 3 if P1 > 4 else 0, P1 - 1):
            row_base = (((m1 >> rbl) * (pitch >> 4)) << (4 + rbl)) + ((m1 & ((1 << rbl) - 1)) << 4)
            tid = np.arange(NT)[:, None]; it = np.arange(R // NT)[None, :]
            lane_off = row_base + ((tid >> 4) << (4 + rbl)) + (tid & 15)
            it_stride = (NT >> 4) << (4 + rbl)
            assert np.array_equal(lane_off + it * it_stride, t2_off(m1, tid + it * NT, pitch, rb))
            assert row_base + ((R >> 4) << (4 + rbl)) == t2_off(m1, R, pitch, rb)         # the Nyquist element


@pytest.mark.parametrize('L,R', [(1, 32), (2, 32), (16, 32), (17, 32), (31, 32), (32, 32), (100, 128), (128, 128),
                                 (3000, 4096), (4096, 4096)])
def test_lag_validity_is_one_range_test(L, R):
    P2 = 2 * R
    u = np.arange(P2)
    j = (u + L - 1) % P2
    ok_by_column = j < 2 * L - 1
    span = np.uint32(P2 - 2 * L)
    ok_by_range = (u - L).astype(np.uint32) > span            # unsigned: u < L wraps to a huge number
    assert np.array_equal(ok_by_column, ok_by_range)
    assert ok_by_column.sum() == 2 * L - 1 and np.array_equal(np.sort(j[ok_by_column]), np.arange(2 * L - 1))
    # the odd output of a pair (u0 + 1 with u0 = u - L as uint32, possibly 0xffffffff)
    u0 = (u[::2] - L).astype(np.uint32)
    assert np.array_equal((u0 + np.uint32(1)) > span, ok_by_column[1::2])
