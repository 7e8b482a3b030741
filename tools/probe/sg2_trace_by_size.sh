#!/bin/bash
# per-(kernel, grid size) durations of the StyleGAN2 256 x 256 iteration (GPU box): which SIZES of the bandwidth-bound kernels the
# time goes to -> gpurun_out/r6/sg2_by_size.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg2kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/sg2kt -- python3 $R/bench.py --workload stylegan2 --steps 4 --warmup 2 --no_cpu_baseline > $OUT/sg2kt.log 2>&1
cd $R
python - <<'PY' > $OUT/sg2_by_size.txt
import csv, glob, collections
f = glob.glob('/tmp/sg2kt/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'conv_gemm' in n or 'conv_wgrad' in n or 'conv_wino' in n:
        continue
    key = (n[:90], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0))))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = 6.0
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{'kernel':90s} {'grid':>10s} {'wg':>5s} {'calls/it':>8s} {'ms/it':>8s} {'us/call':>9s}")
for (n, g, w), (c, us) in rows[:110]:
    print(f"{n:90s} {g:10d} {w:5d} {c / steps:8.1f} {us / steps / 1e3:8.3f} {us / c:9.1f}")
PY
head -5 $OUT/sg2_by_size.txt
