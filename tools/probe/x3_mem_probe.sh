tools/x3_variants.sh fp32 base x3abl36 x3abl100 x3abl128 x3abl132
echo "fp32 kernel with the blocked-layout addresses:"
DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_x3abl128.so python tools/wino4x_time.py --fp32 2>&1 | tail -1
