"""Where the GPU ball-tree builder spends its time, phase by phase (s_memtime stamps of the first workgroup's first
thread, printed by the kernel).  Needs a diagnostic library:
    cd kerneldensityestimate.jl_amd/csrc && hipcc <the Makefile's flags> -DKDEHIP_TREE_STAMPS -c treebuild.hip -o /tmp/tb.o \\
        && hipcc -shared -fPIC --offload-arch=gfx950 -o ../libkdehip_ts.so $(ls build/*.o | grep -v treebuild.o) /tmp/tb.o
End of round 2, 6 x 2048 points, warm: quick-select 48 % (one wavefront per range: latency bound at the top depths),
widest-dimension sums 21 % (the reference's sequential sums: one lane per range and dimension), bottom-up statistics
17 % (through global memory, one round of dependent loads per depth), gather of the points in leaf order 6 %, keys 4 %."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDEHIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kerneldensityestimate.jl_amd", "libkdehip_ts.so")
import numpy as np
import kdehip
rng = np.random.default_rng(0)
for D, N in ((6, 2048), (3, 2000)):
    p = rng.standard_normal((D, N)); k = np.full(D, 0.3)
    for _ in range(3): kdehip.kde(p, k, device=0)
