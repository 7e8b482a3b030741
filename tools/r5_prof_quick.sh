#!/bin/bash
# quick look: kernel-trace stats of the default workload + the ATen operators of a step (GPU box)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt32 -- python3 $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg > $OUT/kt32.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cd $R
python tools/sngan_aten_ops.py sngan32 2>&1 | grep -v amdgpu.ids | tee $OUT/aten32.txt | head -40
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan eager', d['value'], d['ms_per_step'])"
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan graph', d['value'], d['ms_per_step'])"
DIAGAN_WGRAD_BATCH=0 DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan graph, no wgrad batch', d['value'], d['ms_per_step'])"
