# run tools/wino4_time.py for every library variant under gpurun_variants/ (GPU box)
mkdir -p gpurun_out/r3c
for v in "$@"; do
  echo "=== $v"
  DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so W4_SHORT=1 timeout 300 python tools/wino4_time.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3c/variants_$(date +%H%M%S).txt
