# round 5 (c): screened step v2 (candidates requested ahead, tighter bound, header loads hoisted) vs v1; stamps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_nokept.so $L/libkdehip_v2.so --configs c3 --rounds 9 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
KDEHIP_LIB=$GRAFT_REPO_ROOT/$L/libkdehip_stamps.so python scripts/screen_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
KDEHIP_LIB=$GRAFT_REPO_ROOT/$L/libkdehip_v2.so python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/stats.txt
import sys; sys.path.insert(0, '.')
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
plan.sample(Nout, Niter=Niter, seed=20260101)
print("screen stats", plan.screen_stats())
PY
