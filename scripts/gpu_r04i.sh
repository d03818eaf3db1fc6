# round 4 (i): A/B of the one-constant exp reduction; the GPU suite on the fraction-tree build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
bash scripts/gpu_ab.sh "b3 e3" c3 > gpurun_out/r04i/ab_c3.txt 2>&1
tail -n 4 gpurun_out/r04i/ab_c3.txt
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r04i/gpu_tests.txt 2>&1
tail -n 8 gpurun_out/r04i/gpu_tests.txt
