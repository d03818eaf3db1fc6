# julia_crosscheck.jl -- the definitive parity check of the oracle against the REAL reference.
#
# NOT RUN in the build image (Julia is absent; DESIGN.md "Oracle": Gibbs parity unpinned).  A maintainer with Julia and
# KernelDensityEstimate.jl installed runs, from the repo root:
#
#     python tests/golden/make_gibbs_kat.py --dump-text /tmp/kat      # inputs + the oracle's outputs, as text
#     julia oracle/julia_crosscheck.jl /tmp/kat/c1 /tmp/kat/c2 /tmp/kat/d6 /tmp/kat/c3      # c3 = the headline shape
#
#     julia oracle/julia_crosscheck.jl --shim /tmp/kat/c1 /tmp/kat/c2 /tmp/kat/d6 /tmp/kat/c3   # on a box with an MI355X:
#         additionally loads kerneldensityestimate.jl_amd/julia/KernelDensityEstimateHIP.jl and checks (a) that the trees
#         its `kde!(points, ks)` builds (kdehip_make_density: what enable!() installs in place of the reference's
#         constructor) are the reference's arrays BIT FOR BIT, `next` included, (b) that the GPU product on the same
#         streams returns the reference's labels and points, and (c) that `mul_batch` (many `*` in one call) returns the
#         densities of the single calls
#
# Each directory holds meta.txt (D M N Np Niter), points_j.txt (D x N), bw_j.txt (D), randU.txt, randN.txt and the
# oracle's indices.txt (M x Np) and pGM.txt (D x Np).  The same random streams are handed to the reference through
# its own keywords (src/MSGibbs01.jl:661-662), so labels must be identical and points equal to ~1e-12.
using KernelDensityEstimate, DelimitedFiles

const WITH_SHIM = "--shim" in ARGS
if WITH_SHIM
  include(joinpath(@__DIR__, "..", "kerneldensityestimate.jl_amd", "julia", "KernelDensityEstimateHIP.jl"))
end

# every array (and `next`) of a tree the shim built against the reference's own, bit for bit
function same_tree(a::BallTreeDensity, b::BallTreeDensity)
  ok = a.bt.dims == b.bt.dims && a.bt.num_points == b.bt.num_points && a.bt.next == b.bt.next &&
       a.multibandwidth == b.multibandwidth
  for f in (:centers, :ranges, :weights, :left_child, :right_child, :lowest_leaf, :highest_leaf, :permutation)
    ok &= getfield(a.bt, f) == getfield(b.bt, f)
  end
  for f in (:means, :bandwidth, :bandwidthMin, :bandwidthMax)
    ok &= getfield(a, f) == getfield(b, f)
  end
  return ok
end

function crosscheck(dir::String)
  D, M, N, Np, Niter = Int.(vec(readdlm(joinpath(dir, "meta.txt"))))
  trees = BallTreeDensity[]
  for j in 1:M
    pts = reshape(Float64.(readdlm(joinpath(dir, "points_$j.txt"))), D, N)
    bw = Float64.(vec(readdlm(joinpath(dir, "bw_$j.txt"))))
    push!(trees, kde!(pts, bw))
  end
  randU = Float64.(vec(readdlm(joinpath(dir, "randU.txt"))))
  randN = Float64.(vec(readdlm(joinpath(dir, "randN.txt"))))
  dummy = kde!(zeros(D, Np) .+ reshape(collect(1.0:Np), 1, Np), [1.0])     # only Npts(dummy) is read (:658)
  pGM, idx = prodAppxMSGibbsS(dummy, trees, nothing, nothing; Niter=Niter, randU=randU, randN=randN)
  want_idx = reshape(Int.(readdlm(joinpath(dir, "indices.txt"))), M, Np)
  want_pts = reshape(Float64.(readdlm(joinpath(dir, "pGM.txt"))), D, Np)
  bad = count(idx .!= want_idx)
  dx = maximum(abs.(pGM .- want_pts))
  println(dir, ": label mismatches = ", bad, " of ", length(idx), ", max |dx| = ", dx)
  ok = bad == 0 && dx < 1e-10
  if WITH_SHIM
    H = KernelDensityEstimateHIP
    for j in 1:M
      pts = reshape(Float64.(readdlm(joinpath(dir, "points_$j.txt"))), D, N)
      bw = Float64.(vec(readdlm(joinpath(dir, "bw_$j.txt"))))
      t = same_tree(H.kde!(pts, bw, nothing), trees[j])
      println(dir, ": tree ", j, " built by kdehip_make_density == the reference's: ", t)
      ok &= t
    end
    # many `*` in one call (kdehip_mul_device_batch) against the same `*` one at a time (kdehip_mul_device): every array
    dd = [H.DeviceDensity(t) for t in trees]
    prods = [dd[1:min(2, M)], dd, dd[1:1]]
    sds = UInt64[11, 12, 13]
    batch = H.mul_batch(prods; seeds=sds)
    for (k, p) in enumerate(prods)
      one = *(p; seed=sds[k])
      t = same_tree(BallTreeDensity(batch[k]), BallTreeDensity(one))
      println(dir, ": batched `*` ", k, " == the single call: ", t)
      ok &= t
    end
    gp, gi = H.prodAppxMSGibbsS(dummy, trees, nothing, nothing; Niter=Niter, randU=copy(randU), randN=copy(randN))
    gbad, gdx = count(gi .!= idx), maximum(abs.(gp .- pGM))
    println(dir, ": GPU vs the reference itself: label mismatches = ", gbad, ", max |dx| = ", gdx)
    ok &= gbad == 0 && gdx < 1e-10
  end
  return ok
end

ok = all(crosscheck.(filter(a -> a != "--shim", ARGS)))
println(ok ? "PARITY WITH THE REFERENCE: OK" : "PARITY WITH THE REFERENCE: FAILED")
exit(ok ? 0 : 1)
