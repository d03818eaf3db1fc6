"""kdehip -- MI355X-native multiscale-Gibbs KDE products (hot path of KernelDensityEstimate.jl).

This package is the Python host mirror of the reference's interface for that path
(`kde!`, `BallTreeDensity`, `getPoints/getBW/getWeights`, `Npts/Ndim`, `prodAppxMSGibbsS`, `gibbs1`)
over the C ABI of libkdehip.so (include/kdehip.h).  The directory name contains a dot, so import it
through the top-level `kdehip` module of this repository.
"""
from ._lib import KdeHipError, LIB_PATH, lib as _clib  # noqa: F401  (import fails loudly if the .so is missing)
from .density import (BallTree, BallTreeDensity, Ndim, Npts, density_from_arrays, getBW, getPoints,  # noqa: F401
                      getWeights, kde, kde_b, kde_batch)
from .bandwidth import auto_bandwidth, evaluateDualTree, kde_auto  # noqa: F401
from .product import (DeviceDensity, GbGlb, MultiProductPlan, ProductBatch, ProductPlan, gibbs1, makeEmptyGbGlb, mul, mul_device, mul_device_batch,  # noqa: F401
                      nlevels, philox_streams, prodAppxMSGibbsS, prodAppxMSGibbsS_batch, prodAppxMSGibbsS_device,
                      prodAppxMSGibbsS_resident)


def device_count() -> int:
    return int(_clib.kdehip_device_count())


def version() -> int:
    return int(_clib.kdehip_version())
