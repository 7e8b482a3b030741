# final measurements of round 3 (GPU box): default bench (the scored command), the other workloads, a bare 2-rank launch,
# then tools/profile_round.sh -> gpurun_out/prof_r03/
O=gpurun_out/r3final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload sngan64 --no_cpu_baseline > $O/bench64.json 2> $O/bench64.err
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg > $O/bench_p2.json 2> $O/bench_p2.err
python bench.py --workload sngan64 --phase 2 --no_cpu_baseline > $O/bench64_p2.json 2> $O/bench64_p2.err
python bench.py --workload dcgan --no_cpu_baseline > $O/bench_dcgan.json 2> $O/bench_dcgan.err
python bench.py --workload stylegan2 --no_cpu_baseline > $O/bench_sg2.json 2> $O/bench_sg2.err
DIAGAN_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
rm -rf gpurun_out/prof_r03
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
for f in $O/bench*.json; do echo $f; python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['unit'], d['ms_per_step'], d['config'].get('workload'), d.get('roofline',{}).get('frac'))
"; done
