# round 5 (w): kernel-level view of the bandwidth search (rocprofv3 --kernel-trace --stats): speculative rounds at 6 x 1000,
# plain rounds at 6 x 2048
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05w; mkdir -p $O
REPO=$PWD
cat > /tmp/loocv_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, kdehip
from tests.helpers import synth_mixture
N = int(sys.argv[1])
pts = synth_mixture(np.random.default_rng(N), 6, N)
for _ in range(40):
    bw, ne = kdehip.auto_bandwidth(pts, return_evals=True)
print(N, ne, bw[0])
PY
cd /tmp && export TMPDIR=/tmp
for N in 1000 2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$N -- python3 /tmp/loocv_one.py $N > $O/run_$N.txt 2> $O/err_$N.txt
done
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05w")
for N in (1000, 2048):
    f = max(glob.glob(f"{O}/n{N}/**/*kernel_stats.csv", recursive=True), key=os.path.getsize)
    print(f"== 6 x {N}: 40 searches")
    for row in csv.DictReader(open(f)):
        print("  %-70s calls %6s  avg %9.1f ns  total %10.0f ns  %5s %%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]), float(row["TotalDurationNs"]), row["Percentage"]))
PY
find $O -type f -size +2M -delete
