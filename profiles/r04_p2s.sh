mkdir -p gpurun_out/r4i
for w in 0 5; do echo "== wave $w"; GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_p2s$w.so python3 profiles/p2_stamps.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i/p2_stamps.txt
