# julia_crosscheck.jl -- the definitive parity check of the oracle against the REAL reference.
#
# NOT RUN in the build image (Julia is absent; DESIGN.md "Oracle": Gibbs parity unpinned).  A maintainer with Julia and
# KernelDensityEstimate.jl installed runs, from the repo root:
#
#     python tests/golden/make_gibbs_kat.py --dump-text /tmp/kat      # inputs + the oracle's outputs, as text
#     julia oracle/julia_crosscheck.jl /tmp/kat/c1 /tmp/kat/c2 /tmp/kat/d6 /tmp/kat/c3      # c3 = the headline shape
#
# Each directory holds meta.txt (D M N Np Niter), points_j.txt (D x N), bw_j.txt (D), randU.txt, randN.txt and the
# oracle's indices.txt (M x Np) and pGM.txt (D x Np).  The same random streams are handed to the reference through
# its own keywords (src/MSGibbs01.jl:661-662), so labels must be identical and points equal to ~1e-12.
using KernelDensityEstimate, DelimitedFiles

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
  return bad == 0 && dx < 1e-10
end

ok = all(crosscheck.(ARGS))
println(ok ? "PARITY WITH THE REFERENCE: OK" : "PARITY WITH THE REFERENCE: FAILED")
exit(ok ? 0 : 1)
