import subprocess,sys,re
src,D=sys.argv[1],sys.argv[2]
cmd=f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -disable-vector-combine -DKDEHIP_DIM={D} {' '.join(sys.argv[3:])} -c gibbs_kernel.hip -o /tmp/ru_{D}.o -Rpass-analysis=kernel-resource-usage 2>&1"
out=subprocess.run(cmd,shell=True,capture_output=True,text=True).stdout
cur=None
for line in out.splitlines():
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        n=m.group(1)
        mm=re.search(r"gibbs_product_kernelI(\w)Li(\d)ELi(\d)ELi(\d+)E",n)
        cur=f"{ {'d':'f64','f':'f32'}[mm.group(1)]} D{mm.group(2)} mode{mm.group(3)} W{mm.group(4)}" if mm else n[:40]
        vals={}
        continue
    for key in ("TotalSGPRs","VGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs Spill","VGPRs Spill","LDS Size [bytes/block]"):
        m=re.search(re.escape(key)+r": (\d+)",line)
        if m and cur:
            vals[key]=m.group(1)
            if key.startswith("LDS Size"):
                print(cur, "sgpr",vals.get("TotalSGPRs"),"vgpr",vals.get("VGPRs"),"scratch",vals.get("ScratchSize [bytes/lane]"),"occ",vals.get("Occupancy [waves/SIMD]"),"sspill",vals.get("SGPRs Spill"),"vspill",vals.get("VGPRs Spill"))
