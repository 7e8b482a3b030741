mkdir -p gpurun_out/r3c
timeout 300 python tools/wino4_check.py --quick 2>&1 | grep -v amdgpu.ids | grep -E "WORST|float64" 
W4_SHORT=1 timeout 300 python tools/wino4_time.py 2>&1 | grep -v amdgpu.ids
echo "=== nt stores"
DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_w4_nt.so W4_SHORT=1 timeout 300 python tools/wino4_time.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/wino4_ksweep.py 2>&1 | grep -v amdgpu.ids | grep "cfg 13"
