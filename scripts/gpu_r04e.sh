cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
for T in 0 1 auto; do
  for cfg in "c2 64 0" "c2 8 0" "c3 8 256" "c3 4 2048" "c3 16 512"; do
    set -- $cfg
    if [ "$T" = "auto" ]; then unset KDEHIP_BATCH_TABLES; else export KDEHIP_BATCH_TABLES=$T; fi
    python bench.py --config $1 --batch $2 --nout $3 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tables=$T', '$1 x $2 nout $3:', 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'speedup', round(d['back_to_back']['batched_speedup'],2), 'same', d['batched_equals_single_calls_bit_for_bit'])"
  done
done > $O/batch_tables.txt 2>&1
cat $O/batch_tables.txt
python - <<'PY' > $O/hipinit.txt 2>&1
import ctypes, time
t0=time.perf_counter(); h=ctypes.CDLL("libamdhip64.so"); t1=time.perf_counter()
n=ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); t2=time.perf_counter()
h.hipSetDevice(0); h.hipFree(None); t3=time.perf_counter()
print("no libkdehip: dlopen libamdhip64 %.1f ms, hipGetDeviceCount %.1f ms, context %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
PY
python - <<'PY' >> $O/hipinit.txt 2>&1
import ctypes, time, os
t0=time.perf_counter(); k=ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"kerneldensityestimate.jl_amd/libkdehip.so")); t1=time.perf_counter()
h=ctypes.CDLL("libamdhip64.so")
n=ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); t2=time.perf_counter()
h.hipSetDevice(0); h.hipFree(None); t3=time.perf_counter()
print("with libkdehip: dlopen libkdehip %.1f ms, hipGetDeviceCount %.1f ms, context %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
PY
python - <<'PY' >> $O/hipinit.txt 2>&1
import ctypes, time
t0=time.perf_counter(); h=ctypes.CDLL("libamdhip64.so"); t1=time.perf_counter()
n=ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); t2=time.perf_counter()
print("again, no libkdehip: hipGetDeviceCount %.1f ms" % ((t2-t1)*1e3))
PY
cat $O/hipinit.txt
