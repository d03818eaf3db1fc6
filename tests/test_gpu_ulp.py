"""The hardware premise of the fp32 screen's error bound, measured (csrc/screen_device.hpp, csrc/selftest.hip).

`ScreenConst` budgets v_rcp_f32, v_rsq_f32 and v_exp_f32 at "1 ulp" = a relative error of at most 2 u (u = 2^-24) each.
Nothing in the reference pins that (src/MSGibbs01.jl:250-351 is fp64; the screen only has to reproduce its DECISIONS),
so the device is asked: every fp32 input the screen can feed the instruction -- and far beyond -- goes through it and is
compared with fp64 arithmetic on the device (kdehip_selftest_fp32).  The sweeps together cover 2^32 bit patterns'
worth of inputs in well under a second of GPU time."""
import ctypes as C
import struct

import pytest

import kdehip
from kdehip import _lib

pytestmark = pytest.mark.gpu

U_BUDGET = 2.0   # units of u = 2^-24: what screen_device.hpp's kx / vc constants charge per approximate instruction


def _bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _float(b: int) -> float:
    return struct.unpack("<f", struct.pack("<I", b))[0]


def _sweep(which, first, last):
    """largest error over the bit patterns first..last (inclusive), the worst input and the hardware's result for it"""
    err, wb, wr = C.c_double(), C.c_uint32(), C.c_uint32()
    _lib.check(_lib.lib.kdehip_selftest_fp32(which, first, last - first + 1, 0, C.byref(err), C.byref(wb), C.byref(wr)))
    return err.value, _float(wb.value), _float(wr.value)


@pytest.mark.parametrize("name,which,lo,hi,why", [
    # v_rcp_f32: the screen takes 1 / (cmin + cov) with both terms in [2^-7, 2^8]; swept: every input whose reciprocal is a
    # normal number, both signs
    ("v_rcp_f32 +", 0, _bits(2.0 ** -126), _bits(2.0 ** 126), "1 / c_d"),
    ("v_rcp_f32 -", 0, _bits(-2.0 ** -126), _bits(-2.0 ** 126), "symmetry"),
    # v_rsq_f32: of prod_d c_d, at most 8 factors in [2^-7, 2^9]; swept: every positive normal number
    ("v_rsq_f32", 1, _bits(2.0 ** -126), 0x7F7FFFFF, "rsqrt(prod c_d)"),
    # v_exp_f32 (2^x): the screen's exponents are <= 0; swept: -0 and every negative input down to -126 (normal results),
    # and the non-negative ones up to 127 for completeness
    ("v_exp_f32 x<=0", 2, 0x80000000, _bits(-126.0), "2^x, x in [-126, -0]"),
    ("v_exp_f32 x>=0", 2, 0x00000000, _bits(127.0), "2^x, x in [0, 127]"),
])
def test_hardware_fp32_approximations_are_within_the_screens_budget(name, which, lo, hi, why):
    err, x, r = _sweep(which, lo, hi)
    print(f"{name}: {hi - lo + 1} inputs, max relative error {err:.4f} u at x = {x!r} (hardware: {r!r})")
    assert err <= U_BUDGET, f"{name} ({why}): {err} u at x = {x!r} exceeds the {U_BUDGET} u ScreenConst budgets"


def test_exp2_below_the_normal_range_is_off_by_less_than_the_smallest_normal():
    """x < -126 (down to -inf): whatever the hardware returns -- zero, a denormal -- is within 2^-126 of 2^x, which is what
    the bound's flush-to-zero term assumes (screen_device.hpp: 'Terms that fp32 flushes to zero ...')."""
    err, x, r = _sweep(3, _bits(-126.0) + 1, 0xFF800000)
    print(f"v_exp_f32 x<-126: max |result - 2^x| = {err:.4f} x 2^-126 at x = {x!r} (hardware: {r!r})")
    assert err <= 1.0


def test_the_sweeps_cover_more_than_the_screens_range_checks_admit():
    """the ranges above contain everything kScreen{MinVar,MaxVar} (csrc/kdehip_internal.hpp) lets through: c_d = tile
    variance + leave-one-out variance in [2^-7, 2^9], products of up to 8 of them in [2^-56, 2^72]"""
    assert 2.0 ** -126 <= 2.0 ** -7 and 2.0 ** 9 <= 2.0 ** 126
    assert _float(0x7F7FFFFF) > 2.0 ** 72 and 2.0 ** -56 > 2.0 ** -126
    assert _lib.MAX_DIMS == 8
