# round 4 (h): A/B of the fraction-tree FAST evaluator + the team tests on a TEAMS=1 build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
bash scripts/gpu_ab.sh "b3 f3" c3 > gpurun_out/r04h/ab_c3.txt 2>&1
bash scripts/gpu_ab.sh "b5 f5" c5 "--rounds 5 --steps 4" > gpurun_out/r04h/ab_c5.txt 2>&1
bash scripts/gpu_ab.sh "b4 f4" c4 "--rounds 5 --steps 6" > gpurun_out/r04h/ab_c4.txt 2>&1
KDEHIP_LIB=$GRAFT_REPO_ROOT/kerneldensityestimate.jl_amd/libkdehip_teams.so timeout 900 python -m pytest tests/test_gpu_team.py -q -m gpu -x > gpurun_out/r04h/team_tests.txt 2>&1
tail -n 12 gpurun_out/r04h/ab_c3.txt gpurun_out/r04h/ab_c5.txt gpurun_out/r04h/ab_c4.txt
tail -n 5 gpurun_out/r04h/team_tests.txt
