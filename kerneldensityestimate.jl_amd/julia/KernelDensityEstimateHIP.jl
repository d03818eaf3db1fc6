# KernelDensityEstimateHIP.jl -- thin `ccall` shim over libkdehip.so (include/kdehip.h).
#
# Host code stays in Julia: densities are built by the reference's own `kde!` / `BallTreeDensity`
# (KernelDensityEstimate.jl), and only the Gibbs engine `gibbs1` (reference src/MSGibbs01.jl:527-629)
# is replaced by the MI355X HIP kernels.  Two ways to use it:
#
#   using KernelDensityEstimate, KernelDensityEstimateHIP
#   pGM, idx = KernelDensityEstimateHIP.prodAppxMSGibbsS(dummy, [p1; p2; p3], nothing, nothing; Niter=5)
#
# or, to route every existing caller (`*`, IncrementalInference, ...) through the GPU:
#
#   KernelDensityEstimateHIP.enable!()      # overrides KernelDensityEstimate.prodAppxMSGibbsS and .gibbs1 (the hot path)
#   KernelDensityEstimateHIP.enable!(kde=true, trees=true, evaluate=true)
#                                           # opt-in: also kde!(points), kde!(points, ks[, weights]) and evaluateDualTree,
#                                           # i.e. the WHOLE `*` -- product, bandwidth search AND tree construction
#
# After enable!() a call of the reference's `prodAppxMSGibbsS` WITHOUT `randU=`/`randN=` (what `*` and every
# JuliaRobotics caller does) no longer draws `rand(Np*Ndens*(Niter+2)*Nlevels)` / `randn(...)` on the host
# (src/MSGibbs01.jl:661-662; 9.4 MB at 6-D x 4 x 1000, Np = 2048) nor uploads them: it goes to
# `kdehip_prod_philox`, whose random numbers are drawn on the device.  Calls WITH explicit streams keep the
# reference's consumption order through the `gibbs1` override.
#
# NOTE: Julia is not installed in the build container, so this file has been written against the
# C ABI but never executed; tests/ exercise the same entry points through the Python mirror, and
# tests/test_julia_shim_syntax.py checks block structure, every ccall against include/kdehip.h and the list of
# methods enable!() overrides.
module KernelDensityEstimateHIP

using KernelDensityEstimate
const KDE = KernelDensityEstimate

const libkdehip = get(ENV, "KDEHIP_LIB", joinpath(@__DIR__, "..", "libkdehip.so"))

# struct kdehip_density (include/kdehip.h)
struct CDensity
  npts::Int64
  ndim::Int64
  means::Ptr{Float64}
  bandwidth::Ptr{Float64}
  weights::Ptr{Float64}
  left_child::Ptr{Int64}
  right_child::Ptr{Int64}
  permutation::Ptr{Int64}
end

CDensity(bd::BallTreeDensity) = CDensity(bd.bt.num_points, bd.bt.dims, pointer(bd.means), pointer(bd.bandwidth),
                                         pointer(bd.bt.weights), pointer(bd.bt.left_child),
                                         pointer(bd.bt.right_child), pointer(bd.bt.permutation))

lasterror() = unsafe_string(ccall((:kdehip_last_error, libkdehip), Cstring, ()))
devicecount() = Int(ccall((:kdehip_device_count, libkdehip), Cint, ()))

const KDEHIP_ERR_UNSUPPORTED = -7        # ndims > 8, Ndens > 16, ... (include/kdehip.h)

function check(rc::Integer)
  rc == 0 && return nothing
  msg = lasterror()
  rc == -3 && throw(BoundsError(msg))   # randU / randN too short
  error("libkdehip ($rc): $msg")         # incl. "kdes must have same dimension"
end

# The reference's own gibbs1: `KDE.gibbs1` until enable!() has overwritten that method, the saved original
# afterwards (calling KDE.gibbs1 from the fallback paths would then recurse into this module).
const ORIGINAL_GIBBS1 = Ref{Any}(nothing)
reference_gibbs1(args...; kw...) =
  ORIGINAL_GIBBS1[] === nothing ? KDE.gibbs1(args...; kw...) : ORIGINAL_GIBBS1[](args...; kw...)

# the same for the front end (enable!() overwrites KDE.prodAppxMSGibbsS as well)
const ORIGINAL_PROD = Ref{Any}(nothing)
reference_prodAppxMSGibbsS(args...; kw...) =
  ORIGINAL_PROD[] === nothing ? KDE.prodAppxMSGibbsS(args...; kw...) : ORIGINAL_PROD[](args...; kw...)

# ... and for the callers either side of the product: `kde!(points)` (LOOCV bandwidth; the second half of `*`,
# src/MSGibbs01.jl:725), the explicit-bandwidth constructors `kde!(points, ks[, weights])` (src/KDE01.jl:34-76: every
# tree of every caller) and `evaluateDualTree` (src/DualTree01.jl:370-421)
const ORIGINAL_KDE_AUTO = Ref{Any}(nothing)
reference_kde_auto(args...) = ORIGINAL_KDE_AUTO[] === nothing ? KDE.kde!(args...) : ORIGINAL_KDE_AUTO[](args...)
const ORIGINAL_KDE_BW = Ref{Any}(nothing)     # kde!(points, ks, addop, diffop)
const ORIGINAL_KDE_BWW = Ref{Any}(nothing)    # kde!(points, ks, weights, addop, diffop)
reference_kde_bw(args...) = ORIGINAL_KDE_BW[] === nothing ? KDE.kde!(args...) : ORIGINAL_KDE_BW[](args...)
reference_kde_bww(args...) = ORIGINAL_KDE_BWW[] === nothing ? KDE.kde!(args...) : ORIGINAL_KDE_BWW[](args...)
const ORIGINAL_EVAL = Ref{Any}(nothing)
const ORIGINAL_EVAL_BD = Ref{Any}(nothing)
reference_evaluateDualTree(args...) =
  ORIGINAL_EVAL[] === nothing ? KDE.evaluateDualTree(args...) : ORIGINAL_EVAL[](args...)
reference_evaluateDualTree_bd(args...) =
  ORIGINAL_EVAL_BD[] === nothing ? KDE.evaluateDualTree(args...) : ORIGINAL_EVAL_BD[](args...)

isEuclidOps(addop, diffop) = all(f -> f === +, addop) && all(f -> f === -, diffop)
# the direct evaluation kernel stands for the reference's default only (FORCE_EVAL_DIRECT = true,
# src/KernelDensityEstimate.jl:54; setForceEvalDirect!(false) brings the dual-tree recursion back: reference path)
directEval() = KDE.FORCE_EVAL_DIRECT

isEuclid(addop, diffop, getMu, getLambda) =
  all(f -> f === +, addop) && all(f -> f === -, diffop) &&
  all(f -> f === KDE.getEuclidMu, getMu) && all(f -> f === KDE.getEuclidLambda, getLambda)

maskbytes(partialDimMask, Ndens, ndims) =
  UInt8[partialDimMask[j][d] ? 0x01 : 0x00 for d in 1:ndims, j in 1:Ndens]   # density-major in memory

"""
    gibbs1(Ndens, trees, Np, Niter, pts, ind, randU, randN; kw...)

Drop-in for `KernelDensityEstimate.gibbs1` (same arguments, fills `pts` and `ind` in place).
Non-Euclidean manifold operators cannot cross the C ABI: they fall back to the reference.
"""
function gibbs1(Ndens::Int, trees::Array{BallTreeDensity,1}, Np::Int, Niter::Int,
                pts::Array{Float64,1}, ind::Array{Int}, randU::Array{Float64,1}, randN::Array{Float64,1};
                addop=(+,), diffop=(-,), getMu=(KDE.getEuclidMu,), getLambda=(KDE.getEuclidLambda,),
                glbs=KDE.makeEmptyGbGlb(), addEntropy::Bool=true,
                ndims::Int=maximum(Ndim.(trees)),
                partialDimMask::AbstractVector{<:BitVector}=[ones(Int, ndims) .== 1 for i in 1:Ndens],
                device::Int=0, ngpus::Int=1)
  fallback() = reference_gibbs1(Ndens, trees, Np, Niter, pts, ind, randU, randN; addop=addop, diffop=diffop,
                                getMu=getMu, getLambda=getLambda, glbs=glbs, addEntropy=addEntropy, ndims=ndims,
                                partialDimMask=partialDimMask)
  isEuclid(addop, diffop, getMu, getLambda) || return fallback()
  cds = CDensity[CDensity(t) for t in trees]
  mask = maskbytes(partialDimMask, Ndens, ndims)
  # glbs.recordChoosen (src/MSGibbs01.jl:29-31): the label trace comes back as labels[level, density, sample]
  Nlevels = floor(Int, log(Float64(maximum(Npts.(trees)))) / log(2.0) + 1.0)   # :568
  labels = glbs.recordChoosen ? zeros(Int32, Nlevels, Ndens, Np) : Int32[]
  GC.@preserve trees cds mask labels begin
    # chains split over `ngpus` devices (device .. device+ngpus-1) in contiguous ranges; ngpus = 1: one GPU
    rc = ccall((:kdehip_gibbs1_multi, libkdehip), Cint,
               (Cint, Ptr{CDensity}, Int64, Cint, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64},
                Int64, Cint, Cint, Ptr{UInt8}, Cint, Cint, Ptr{Int32}),
               Ndens, cds, Np, Niter, pts, ind, randU, length(randU), randN, length(randN),
               addEntropy ? 1 : 0, ndims, mask, device, ngpus, glbs.recordChoosen ? pointer(labels) : C_NULL)
  end
  # shapes beyond the compiled limits (ndims > 8, Ndens > 16) stay on the reference path: the caller sees the
  # same behaviour as without this package, only slower (nothing has been written to pts / ind yet)
  rc == KDEHIP_ERR_UNSUPPORTED && return fallback()
  check(rc)
  if glbs.recordChoosen   # same nesting and 1-based keys as :471-472, :575-583, :109-112
    glbs.labelsChoosen = Dict{Int,Dict{Int,Dict{Int,Int}}}()
    for s in 1:Np
      glbs.labelsChoosen[s] = Dict{Int,Dict{Int,Int}}()
      for j in 1:Ndens
        glbs.labelsChoosen[s][j] = Niter > 0 ? Dict{Int,Int}(l => Int(labels[l, j, s]) for l in 1:Nlevels) :
                                               Dict{Int,Int}()
      end
    end
  end
  nothing
end

"""
    prodAppxMSGibbsS(npd0, trees, anFcns, anParams; Niter=3, ..., seed=nothing, device=0, ngpus=1)

Same keywords and return value as the reference (src/MSGibbs01.jl:645-703), `maxNp` and `Nlevels` included
(:659-660; they only size the random streams, the engine derives its level count from the trees, :568).
With `randU`/`randN` given they are consumed exactly like the reference consumes them; otherwise the
on-device Philox stream keyed by `seed` replaces `rand`/`randn`.
"""
function prodAppxMSGibbsS(npd0::BallTreeDensity, trees::Array{BallTreeDensity,1}, anFcns, anParams;
                          Niter::Int=3, addop=(+,), diffop=(-,), getMu=(KDE.getEuclidMu,),
                          getLambda=(KDE.getEuclidLambda,), glbs=KDE.makeEmptyGbGlb(), addEntropy::Bool=true,
                          ndims::Integer=maximum(Ndim.(trees)), Ndens=length(trees), Np=Npts(npd0),
                          maxNp=maximum([Np; Npts.(trees)]),
                          Nlevels=floor(Int, (log(Float64(maxNp)) / log(2.0)) + 1.0),
                          randU=nothing, randN=nothing,
                          partialDimMask::AbstractVector{<:BitVector}=[ones(Int, ndims) .== 1 for i in 1:length(trees)],
                          seed::Union{Nothing,UInt64}=nothing, device::Int=0, ngpus::Int=1)
  # the reference's own front end and engine, with the streams the reference would have drawn itself
  reference() = reference_prodAppxMSGibbsS(npd0, trees, anFcns, anParams; Niter=Niter, addop=addop, diffop=diffop,
                                           getMu=getMu, getLambda=getLambda, glbs=glbs, addEntropy=addEntropy,
                                           ndims=ndims, Ndens=Ndens, Np=Np, maxNp=maxNp, Nlevels=Nlevels,
                                           randU=(randU === nothing ? rand(Int(Np * Ndens * (Niter + 2) * Nlevels)) : randU),
                                           randN=(randN === nothing ? randn(Int(ndims * Np * (Nlevels + 1))) : randN),
                                           partialDimMask=partialDimMask)
  isEuclid(addop, diffop, getMu, getLambda) || return reference()
  points = zeros(ndims * Np)
  indices = ones(Int, Ndens, Np)
  if randU !== nothing || randN !== nothing || glbs.recordChoosen
    if randU === nothing || randN === nothing   # (label traces go through the drop-in: streams from the host twin of the device RNG)
      s = seed === nothing ? rand(UInt64) : seed
      Ltree = floor(Int, log(Float64(maximum(Npts.(trees)))) / log(2.0) + 1.0)
      K, R = Ndens * (1 + Ltree * (Niter + 1)), ndims * (Ltree + 1)
      if randU === nothing
        randU = zeros(Np * K)
        ccall((:kdehip_philox_fill_uniform, libkdehip), Cvoid, (UInt64, Int64, Int64, Int64, Ptr{Float64}), s, 0, Np, K, randU)
      end
      if randN === nothing
        randN = zeros(Np * R)
        ccall((:kdehip_philox_fill_normal, libkdehip), Cvoid, (UInt64, Int64, Int64, Int64, Ptr{Float64}), s, 0, Np, R, randN)
      end
    end
    gibbs1(Ndens, trees, Np, Niter, points, indices, randU, randN; glbs=glbs, addEntropy=addEntropy,
           ndims=Int(ndims), partialDimMask=partialDimMask, device=device, ngpus=ngpus)
    return reshape(points, ndims, Np), indices
  end
  cds = CDensity[CDensity(t) for t in trees]
  mask = maskbytes(partialDimMask, Ndens, ndims)
  s = seed === nothing ? rand(UInt64) : seed
  GC.@preserve trees cds mask begin
    # one-shot: pack, upload, run (device Philox stream keyed by `s`), copy back; chains split over `ngpus` devices
    rc = ccall((:kdehip_prod_philox, libkdehip), Cint,
               (Cint, Ptr{CDensity}, Int64, Cint, Ptr{Float64}, Ptr{Int64}, UInt64, Cint, Cint, Ptr{UInt8}, Cint, Cint,
                Cint, Ptr{Int32}),
               Ndens, cds, Np, Niter, points, indices, s, addEntropy ? 1 : 0, ndims, mask, 64, device, ngpus, C_NULL)
  end
  rc == KDEHIP_ERR_UNSUPPORTED && return reference()   # beyond the compiled limits (nothing has been written yet)
  check(rc)
  return reshape(points, ndims, Np), indices
end

# the reference's deprecated positional form (src/MSGibbs01.jl:632-643)
function prodAppxMSGibbsS(npd0::BallTreeDensity, trees::Array{BallTreeDensity,1}, anFcns, anParams, Niter::Int)
  @warn "prodApproxMSGibbs has new keyword interface, use (..; Niter::Int=5 ) instead"
  prodAppxMSGibbsS(npd0, trees, anFcns, anParams; Niter=Niter)
end

"""
    DeviceDensity(bd; device=0)

A `BallTreeDensity` uploaded ONCE and kept in HBM (`kdehip_density_upload`; include/kdehip.h section 2c).  Products of
such densities -- `prodAppxMSGibbsS(npd0, ::Vector{DeviceDensity}, ...)` -- are laid out into tiles by the GPU: no host
re-layout, no upload per product.  `free!(d)` releases it (also run by the finalizer).
"""
mutable struct DeviceDensity
  handle::Ptr{Cvoid}
  npts::Int
  ndim::Int
  function DeviceDensity(bd::BallTreeDensity; device::Int=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    cd = Ref(CDensity(bd))
    GC.@preserve bd begin
      check(ccall((:kdehip_density_upload, libkdehip), Cint, (Ref{Ptr{Cvoid}}, Ref{CDensity}, Cint), h, cd, device))
    end
    d = new(h[], Npts(bd), Ndim(bd))
    finalizer(free!, d)
    return d
  end
  function DeviceDensity(handle::Ptr{Cvoid}, npts::Int, ndim::Int)   # a density the library built (resident chain)
    d = new(handle, npts, ndim)
    finalizer(free!, d)
    return d
  end
end
# handle -> DeviceDensity for densities the library built itself (the resident chain below)
function DeviceDensity(handle::Ptr{Cvoid})
  d = ccall((:kdehip_density_npts, libkdehip), Int64, (Ptr{Cvoid},), handle)
  k = ccall((:kdehip_density_ndim, libkdehip), Cint, (Ptr{Cvoid},), handle)
  return DeviceDensity(handle, Int(d), Int(k))
end
function free!(d::DeviceDensity)
  if d.handle != C_NULL
    ccall((:kdehip_density_free, libkdehip), Cvoid, (Ptr{Cvoid},), d.handle)
    d.handle = C_NULL
  end
  nothing
end

"""
    *(trees::Vector{DeviceDensity}; addEntropy=true, seed=nothing) -> DeviceDensity

The reference's `*` (src/MSGibbs01.jl:707-726: product with Niter = 5 and Np = round(mean Npts), then `kde!(pGM)`) on
densities that live in HBM, result in HBM (`kdehip_mul_device`): the sample matrix never leaves the device -- the
bandwidth search reads it there, the tree is built from one copy that comes down meanwhile, the new density's block
goes straight back up.  A belief-propagation sweep chains such products without PCIe traffic per message besides that.
`BallTreeDensity(d)` downloads the reference's arrays when the host wants them.
"""
function Base.:*(trees::Vector{DeviceDensity}; addEntropy::Bool=true, seed::Union{Nothing,UInt64}=nothing)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  handles = Ptr{Cvoid}[t.handle for t in trees]
  s = seed === nothing ? rand(UInt64) : seed
  GC.@preserve trees handles begin
    check(ccall((:kdehip_mul_device, libkdehip), Cint,
                (Ref{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}, UInt64, Cint, Ptr{Float64}, Ptr{Int32}),
                h, length(trees), handles, s, addEntropy ? 1 : 0, C_NULL, C_NULL))
  end
  return DeviceDensity(h[])
end
Base.:*(p::DeviceDensity, q::DeviceDensity) = *([p; q])

struct CMulItem                       # struct kdehip_mul_item, include/kdehip.h
  Ndens::Int32
  addEntropy::Int32
  trees::Ptr{Ptr{Cvoid}}
  seed::UInt64
end

"""
    mul_batch(products::Vector{Vector{DeviceDensity}}; addEntropy=true, seeds=nothing) -> Vector{DeviceDensity}

MANY `*` (src/MSGibbs01.jl:707-726) in ONE library call (`kdehip_mul_device_batch`) -- what a belief-propagation sweep
issues, on the reference's own sizes (100-300 points, test/runtests.jl:189-201): batched sampler, the LOOCV bandwidth
searches of all results of one size in the same launches, the trees on the library's host pool under them.  Result `i` is
bit for bit `*(products[i]; addEntropy, seed=seeds[i])`.
"""
function mul_batch(products::Vector{Vector{DeviceDensity}}; addEntropy::Bool=true,
                   seeds::Union{Nothing,Vector{UInt64}}=nothing)
  n = length(products)
  n == 0 && return DeviceDensity[]
  sds = seeds === nothing ? rand(UInt64, n) : seeds
  handles = [Ptr{Cvoid}[t.handle for t in p] for p in products]
  out = fill(Ptr{Cvoid}(C_NULL), n)
  GC.@preserve products handles begin
    items = CMulItem[CMulItem(length(handles[i]), addEntropy ? 1 : 0, pointer(handles[i]), sds[i]) for i in 1:n]
    check(ccall((:kdehip_mul_device_batch, libkdehip), Cint,
                (Cint, Ptr{CMulItem}, Ptr{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Int32}),
                n, items, out, C_NULL, C_NULL))
  end
  return DeviceDensity[DeviceDensity(h) for h in out]
end

"""
    BallTreeDensity(d::DeviceDensity)

The reference's struct for a density that was BUILT on the device (`*` above): `kdehip_density_download` returns the
twelve arrays of `BallTreeDensity` / `BallTree` (src/BallTreeDensity01.jl:11-24, src/BallTree01.jl:10-28).
"""
function KDE.BallTreeDensity(d::DeviceDensity)
  N, D = d.npts, d.ndim
  centers, ranges, means, bandwidth = zeros(2N * D), zeros(2N * D), zeros(2N * D), zeros(2N * D)
  bwmin, bwmax, weights = zeros(N * D), zeros(N * D), zeros(2N)
  left, right, lowest, highest, perm = zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N)
  check(ccall((:kdehip_density_download, libkdehip), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
               Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              d.handle, centers, ranges, weights, left, right, lowest, highest, perm, means, bandwidth, bwmin, bwmax, C_NULL))
  return density_from_arrays(D, N, centers, ranges, weights, left, right, lowest, highest, perm, means, bandwidth, bwmin, bwmax)
end

# The reference's struct around twelve arrays the library filled: the field order of the reference's constructors
# (src/BallTree01.jl:453-457, src/BallTreeDensity01.jl:225-226), the function handles makeBallTreeDensity installs
# (src/BallTreeDensity01.jl:200-201), uniform bandwidth (multibandwidth = 0, :213), and `next` where buildTree! leaves it:
# it starts at 2 and grows by one per internal node below the root (src/BallTree01.jl:384-393,430) = max(N, 2).
function density_from_arrays(D::Int, N::Int, centers, ranges, weights, left, right, lowest, highest, perm, means, bandwidth,
                             bwmin, bwmax)
  bt = KDE.BallTree(D, N, centers, ranges, weights, left, right, lowest, highest, perm, max(N, 2), KDE.swapDensity!,
                    KDE.calcStatsDensity!, [])
  bd = KDE.BallTreeDensity(bt, KDE.GaussianKer, 0, means, bandwidth, bwmin, bwmax, bt.calcStatsHandle, bt.swapHandle)
  bd.bt.data = bd   # the circular reference the reference keeps "for emulating polymorphism"
  return bd
end

# whether `kde!(points, ks[, weights])` can be built by the library: Euclidean operators, uniform bandwidth given as 1 or D
# standard deviations, at most 8 dimensions, at least one point (everything else: the reference's own constructor, with
# the reference's own errors)
builds_here(points, ks, addop, diffop) =
  isEuclidOps(addop, diffop) && 1 <= size(points, 1) <= 8 && size(points, 2) >= 1 && (length(ks) == 1 || length(ks) == size(points, 1))

"""
    kde!(points, ks[, weights])

`kde!(points, ks, weights)` / `kde!(points, ks)` (src/KDE01.jl:34-76 -> makeBallTreeDensity, src/BallTreeDensity01.jl:192-231
-> buildTree!, src/BallTree01.jl:415-434) built by the library's pooled host builder (`kdehip_make_density`): the same
twelve arrays as the reference's single-threaded quick-select gives (same split rule, swap order, node numbering and
moment matching; pinned by the reference's own golden files) -- bit for bit with unit weights (`kde!(points, ks)`).  With
other weights the normalisation `weights ./ sum(weights)` (src/KDE01.jl:46) is a sequential sum in the library and a
pairwise `@simd` sum in Julia: the total, hence every weight and moment-matched node, may differ in the last bit (which
is why the override `enable!(trees=true)` installs hands only unit weights to the library).
"""
function kde!(points::AbstractMatrix{<:Real}, ks::Vector{Float64}, weights::Union{Nothing,Vector{Float64}}=nothing)
  D, N = size(points)
  pts = Matrix{Float64}(points)
  weights === nothing || length(weights) == N || error("weights must have one entry per point")
  centers, ranges, means, bandwidth = zeros(2N * D), zeros(2N * D), zeros(2N * D), zeros(2N * D)
  bwmin, bwmax, w = zeros(N * D), zeros(N * D), zeros(2N)
  left, right, lowest, highest, perm = zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N)
  GC.@preserve pts ks weights begin
    check(ccall((:kdehip_make_density, libkdehip), Cint,
                (Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}),
                D, N, pts, ks, length(ks), weights === nothing ? C_NULL : pointer(weights), centers, ranges, w, left, right,
                lowest, highest, perm, means, bandwidth, bwmin, bwmax))
  end
  return density_from_arrays(D, N, centers, ranges, w, left, right, lowest, highest, perm, means, bandwidth, bwmin, bwmax)
end

function prodAppxMSGibbsS(npd0, trees::Vector{DeviceDensity}, anFcns, anParams;
                          Niter::Int=3, addEntropy::Bool=true, ndims::Integer=maximum(t.ndim for t in trees),
                          Ndens=length(trees), Np=(npd0 isa BallTreeDensity ? Npts(npd0) : Int(npd0)),
                          partialDimMask::AbstractVector{<:BitVector}=[ones(Int, ndims) .== 1 for i in 1:length(trees)],
                          seed::Union{Nothing,UInt64}=nothing, precision::Int=64)
  points = zeros(ndims * Np)
  indices = ones(Int, Ndens, Np)
  handles = Ptr{Cvoid}[t.handle for t in trees]
  mask = maskbytes(partialDimMask, Ndens, ndims)
  s = seed === nothing ? rand(UInt64) : seed
  GC.@preserve trees handles mask begin
    check(ccall((:kdehip_prod_philox_resident, libkdehip), Cint,
                (Cint, Ptr{Ptr{Cvoid}}, Int64, Cint, UInt64, Cint, Ptr{UInt8}, Cint, Ptr{Float64}, Ptr{Int64}),
                Ndens, handles, Np, Niter, s, addEntropy ? 1 : 0, mask, precision, points, indices))
  end
  return reshape(points, ndims, Np), indices
end

"""
    evaluateDualTree(bd, pos, lvFlag=false)

`evaluateDualTree` / `bd(pos)` with the reference's default `FORCE_EVAL_DIRECT = true`
(src/DualTree01.jl:370-446) on the GPU.  `lvFlag=true`: leave-one-out at `bd`'s own points.
"""
function evaluateDualTree(bd::BallTreeDensity, pos::AbstractMatrix{Float64}, lvFlag::Bool=false; device::Int=0)
  Ndim(bd) == size(pos, 1) || error("bd and pos must have the same dimension")
  Nq = lvFlag ? Npts(bd) : size(pos, 2)
  out = zeros(Nq)
  cd = Ref(CDensity(bd))
  posd = Matrix{Float64}(pos)
  GC.@preserve bd posd begin
    check(ccall((:kdehip_evaluate, libkdehip), Cint, (Ref{CDensity}, Ptr{Float64}, Int64, Cint, Ptr{Float64}, Cint),
                cd, posd, size(pos, 2), lvFlag ? 1 : 0, out, device))
  end
  return out
end

"""
    kde!(points)

`kde!(points)` (src/KDE01.jl:3-27) in ONE library call (`kdehip_make_density_auto`): the per-dimension LOOCV bandwidth is
searched on the GPU while the library's pooled host builder makes the ball tree (topology, bounding boxes, weights and
means do not depend on the bandwidth); the variances are filled in afterwards.  The arrays are those of
`kde!(points, bw)` with the bandwidth found, bit for bit.
"""
function kde!(points::AbstractMatrix{Float64}; device::Int=0)
  D, N = size(points)
  (N < 2 || D > 8) && return reference_kde_auto(points)   # (the library's limits: the reference path)
  bw = zeros(D)
  nev = Ref{Int32}(0)
  pts = Matrix{Float64}(points)
  centers, ranges, means, bandwidth = zeros(2N * D), zeros(2N * D), zeros(2N * D), zeros(2N * D)
  bwmin, bwmax, w = zeros(N * D), zeros(N * D), zeros(2N)
  left, right, lowest, highest, perm = zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N), zeros(Int, 2N)
  check(ccall((:kdehip_make_density_auto, libkdehip), Cint,
              (Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Int32}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
               Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
               Ptr{Float64}),
              D, N, pts, bw, nev, device, centers, ranges, w, left, right, lowest, highest, perm, means, bandwidth, bwmin,
              bwmax))
  return density_from_arrays(D, N, centers, ranges, w, left, right, lowest, highest, perm, means, bandwidth, bwmin, bwmax)
end

"""
    auto_bandwidth(points)

The bandwidth `kde!(points)` selects (D standard deviations; `kdehip_auto_bandwidth`), without building the density.
"""
function auto_bandwidth(points::AbstractMatrix{Float64}; device::Int=0)
  D, N = size(points)
  bw = zeros(D)
  nev = Ref{Int32}(0)
  pts = Matrix{Float64}(points)
  check(ccall((:kdehip_auto_bandwidth, libkdehip), Cint, (Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Int32}, Cint),
              D, N, pts, bw, nev, device))
  return bw
end

"""
    enable!(; kde=false, trees=false, evaluate=false)

Route `KernelDensityEstimate.prodAppxMSGibbsS` and `KernelDensityEstimate.gibbs1` -- and with them `*` and every
downstream caller -- through libkdehip.so (method overwrites).  A product called without `randU`/`randN`
takes the device-RNG one-shot entry (`kdehip_prod_philox`: no host `rand`/`randn`, no upload of the streams);
explicit streams are consumed in the reference's order by the `gibbs1` override.  The reference's own methods
stay reachable for non-Euclidean manifolds and for shapes beyond the compiled limits: they are invoked in the
world age in which they were defined.

The callers either side of the product are OPT-IN by keyword (all off by default: none of this file has been executed
yet -- no Julia in the build image -- so the default keeps the blast radius at the two hot-path methods; switch them on
once `oracle/julia_crosscheck.jl --shim` has passed on your installation): `kde` = `kde!(points)` (LOOCV
bandwidth + tree in one library call), `trees` = `kde!(points, ks)` and `kde!(points, ks, weights)` (every tree of every
caller built by the library's pooled builder instead of the reference's single-threaded quick-select), `evaluate` = both
forms of `evaluateDualTree`.  An override whose reference method cannot be found by its signature (another version of
KernelDensityEstimate.jl) is skipped with a warning: the reference method stays in place.
"""
function enable!(; kde::Bool=false, trees::Bool=false, evaluate::Bool=false)
  devicecount() > 0 || error("libkdehip: no MI355X visible; refusing to enable (no CPU fallback in the library)")
  ORIGINAL_GIBBS1[] === nothing || return nothing   # already enabled
  orig = KDE.gibbs1
  m = first(methods(orig))
  invoke_original(args...; kw...) = Base.invoke_in_world(m.primary_world, orig, args...; kw...)
  origprod = KDE.prodAppxMSGibbsS
  mp = first(mm for mm in methods(origprod) if mm.nargs == 5)   # the keyword method: 4 positional arguments
  invoke_original_prod(args...; kw...) = Base.invoke_in_world(mp.primary_world, origprod, args...; kw...)
  ORIGINAL_GIBBS1[] = invoke_original
  ORIGINAL_PROD[] = invoke_original_prod
  # kde!(points, addop, diffop) (src/KDE01.jl:3-27), kde!(points, ks, addop, diffop) (:64-76), kde!(points, ks, weights,
  # addop, diffop) (:34-57) and the matrix / density forms of evaluateDualTree (src/DualTree01.jl:370-421): found by
  # concrete argument types, invoked in the world they were defined in; a miss leaves the reference method in place
  origkde = KDE.kde!
  origeval = KDE.evaluateDualTree
  Pl, Mi = Tuple{typeof(+)}, Tuple{typeof(-)}
  function saved(f, sig, what)
    try
      mth = which(f, sig)
      return (args...) -> Base.invoke_in_world(mth.primary_world, f, args...)
    catch err
      @warn "KernelDensityEstimateHIP.enable!: reference method not found, override skipped" what err
      return nothing
    end
  end
  kde_auto = kde ? saved(origkde, Tuple{Matrix{Float64},Pl,Mi}, "kde!(points, addop, diffop)") : nothing
  kde_bw = trees ? saved(origkde, Tuple{Matrix{Float64},Vector{Float64},Pl,Mi}, "kde!(points, ks, addop, diffop)") : nothing
  kde_bww = trees ? saved(origkde, Tuple{Matrix{Float64},Vector{Float64},Vector{Float64},Pl,Mi}, "kde!(points, ks, weights, addop, diffop)") : nothing
  eval_m = evaluate ? saved(origeval, Tuple{BallTreeDensity,Matrix{Float64},Bool,Float64,Pl,Mi}, "evaluateDualTree(bd, pos::Matrix)") : nothing
  eval_b = evaluate ? saved(origeval, Tuple{BallTreeDensity,BallTreeDensity,Bool,Float64,Pl,Mi}, "evaluateDualTree(bd, pos::BallTreeDensity)") : nothing
  kde_auto === nothing || (ORIGINAL_KDE_AUTO[] = kde_auto)
  kde_bw === nothing || (ORIGINAL_KDE_BW[] = kde_bw)
  kde_bww === nothing || (ORIGINAL_KDE_BWW[] = kde_bww)
  eval_m === nothing || (ORIGINAL_EVAL[] = eval_m)
  eval_b === nothing || (ORIGINAL_EVAL_BD[] = eval_b)
  @eval KDE function gibbs1(Ndens::Int, trees::Array{BallTreeDensity,1}, Np::Int, Niter::Int,
                            pts::Array{Float64,1}, ind::Array{Int}, randU::Array{Float64,1},
                            randN::Array{Float64,1}; addop=(+,), diffop=(-,), getMu=(getEuclidMu,),
                            getLambda=(getEuclidLambda,), glbs=makeEmptyGbGlb(), addEntropy::Bool=true,
                            ndims::Int=maximum(Ndim.(trees)),
                            partialDimMask::AbstractVector{<:BitVector}=[ones(Int, ndims) .== 1 for i in 1:Ndens])
    if $(isEuclid)(addop, diffop, getMu, getLambda)
      return $(gibbs1)(Ndens, trees, Np, Niter, pts, ind, randU, randN; glbs=glbs, addEntropy=addEntropy,
                       ndims=ndims, partialDimMask=partialDimMask)
    end
    return $(invoke_original)(Ndens, trees, Np, Niter, pts, ind, randU, randN; addop=addop, diffop=diffop,
                              getMu=getMu, getLambda=getLambda, glbs=glbs, addEntropy=addEntropy, ndims=ndims,
                              partialDimMask=partialDimMask)
  end
  # the front end: the reference's keyword list (src/MSGibbs01.jl:645-664) with `nothing` in place of the eager
  # rand(...) / randn(...) defaults, so that "no streams given" can be told from "streams given"
  @eval KDE function prodAppxMSGibbsS(npd0::BallTreeDensity, trees::Array{BallTreeDensity,1}, anFcns, anParams;
                                      Niter::Int=3, addop=(+,), diffop=(-,), getMu=(getEuclidMu,),
                                      getLambda=(getEuclidLambda,), glbs=makeEmptyGbGlb(), addEntropy::Bool=true,
                                      ndims::Integer=maximum(Ndim.(trees)), Ndens=length(trees), Np=Npts(npd0),
                                      maxNp=maximum([Np; Npts.(trees)]),
                                      Nlevels=floor(Int, (log(Float64(maxNp)) / log(2.0)) + 1.0),
                                      randU=nothing, randN=nothing,
                                      partialDimMask::AbstractVector{<:BitVector}=[ones(Int, ndims) .== 1 for i in 1:length(trees)])
    return $(prodAppxMSGibbsS)(npd0, trees, anFcns, anParams; Niter=Niter, addop=addop, diffop=diffop, getMu=getMu,
                               getLambda=getLambda, glbs=glbs, addEntropy=addEntropy, ndims=ndims, Ndens=Ndens,
                               Np=Np, maxNp=maxNp, Nlevels=Nlevels, randU=randU, randN=randN,
                               partialDimMask=partialDimMask)
  end
  # kde!(points) -- the second half of `*` (src/MSGibbs01.jl:725) and of every README usage: LOOCV bandwidth search on the
  # GPU and the tree on the library's host pool, one call (kdehip_make_density_auto)
  kde_auto === nothing || @eval KDE function kde!(points::A, addop::Tuple=(+,), diffop::Tuple=(-,)) where {A <: AbstractArray{Float64,2}}
    if $(isEuclidOps)(addop, diffop) && size(points, 2) >= 2 && size(points, 1) <= 8
      return $(kde!)(points)
    end
    return $(reference_kde_auto)(points, addop, diffop)
  end
  # kde!(points, ks) / kde!(points, ks, weights) -- every tree any caller builds (src/KDE01.jl:34-76): the library's pooled
  # builder (kdehip_make_density); same signatures as the reference's methods, so these definitions replace them
  kde_bw === nothing || @eval KDE function kde!(points::A, ks::Array{Float64,1}, addop::Tuple=(+,), diffop::Tuple=(-,)) where {A <: AbstractArray{Float64,2}}
    if $(builds_here)(points, ks, addop, diffop)
      return $(kde!)(points, ks, nothing)
    end
    return $(reference_kde_bw)(points, ks, addop, diffop)
  end
  kde_bww === nothing || @eval KDE function kde!(points::AbstractArray{<:Real,2}, ks::Array{Float64,1}, weights::Array{Float64,1},
                                                 addop=(+,), diffop=(-,))
    # (unit weights only -- what kde!(points, ks) passes: there `weights ./ sum(weights)` is exact in any summation order)
    if $(builds_here)(points, ks, addop, diffop) && length(weights) == size(points, 2) && all(isone, weights)
      return $(kde!)(points, ks, nothing)
    end
    return $(reference_kde_bww)(points, ks, weights, addop, diffop)
  end
  # evaluateDualTree(bd, pos::Matrix) / bd(pos) / evaluateDualTree(bd, pos::BallTreeDensity): direct evaluation on the GPU
  # while the reference's own default FORCE_EVAL_DIRECT = true stands and the operators are Euclidean
  eval_m === nothing || @eval KDE function evaluateDualTree(bd::BallTreeDensity, pos::Array{Float64,2}, lvFlag::Bool=false, errTol::Float64=1e-3,
                                      addop=(+,), diffop=(-,))
    if $(isEuclidOps)(addop, diffop) && $(directEval)() && bd.bt.dims <= 8 && bd.multibandwidth == 0
      return $(evaluateDualTree)(bd, pos, lvFlag)
    end
    return $(reference_evaluateDualTree)(bd, pos, lvFlag, errTol, addop, diffop)
  end
  eval_b === nothing || @eval KDE function evaluateDualTree(bd::BallTreeDensity, pos::BallTreeDensity, lvFlag::Bool=false, errTol::Float64=1e-3,
                                      addop=(+,), diffop=(-,))
    if $(isEuclidOps)(addop, diffop) && $(directEval)() && bd.bt.dims <= 8 && bd.multibandwidth == 0
      bd.bt.dims == pos.bt.dims || error("bd and pos must have the same dimension")
      return $(evaluateDualTree)(bd, getPoints(pos), lvFlag)
    end
    return $(reference_evaluateDualTree_bd)(bd, pos, lvFlag, errTol, addop, diffop)
  end
  nothing
end

"""
    overridden_methods()

What `enable!(kde=true, trees=true, evaluate=true)` replaces in `KernelDensityEstimate` (pinned by
tests/test_julia_shim_syntax.py; plain `enable!()` replaces the first two): with all of them, an unchanged caller of `*`
runs product, bandwidth search, tree construction and evaluation in libkdehip.so.
"""
overridden_methods() = ["gibbs1", "prodAppxMSGibbsS", "kde!(points)", "kde!(points, ks)", "kde!(points, ks, weights)",
                        "evaluateDualTree(bd, pos::Array{Float64,2})", "evaluateDualTree(bd, pos::BallTreeDensity)"]

end # module
