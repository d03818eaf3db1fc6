#!/usr/bin/env python3
"""Phases of loocv_prep_kernel (a -DKDEHIP_PREP_STAMPS build of csrc/evaluate.hip linked as a development library):
s_memtime of block 0 at the phase boundaries, in shader-clock cycles.
    KDEHIP_LIB=.../libkdehip_prep.so python scripts/prep_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip
from kdehip import _lib
from tests.helpers import synth_mixture
fn = _lib.lib.kdehip_debug_prep_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong)]
for N in (300, 1000, 2048):
    pts = synth_mixture(np.random.default_rng(N), 6, N)
    for _ in range(3):
        kdehip.auto_bandwidth(pts)
    st = (C.c_ulonglong * 8)()
    assert fn(st) == 0
    t = [int(x) for x in st[:5]]
    print(f"6 x {N}: load {t[1]-t[0]} | sort {t[2]-t[1]} | interval arithmetic {t[3]-t[2]} | reduce + init {t[4]-t[3]} | total {t[4]-t[0]} cycles")
