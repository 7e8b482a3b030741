python -m pytest tests/test_conv_gpu.py tests/test_sngan_gpu.py -x -q -k "not float64" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for w in sngan64 sngan32; do
rm -rf /tmp/kt_$w; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg > /tmp/kt_$w.log 2>&1
python3 - $w <<'PY'
import csv,glob,sys
w=sys.argv[1]
f=glob.glob(f"/tmp/kt_{w}/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(w,"total kernel ms/step %.2f"%(tot/9/1e6))
for r in rows:
    if 'finish' in r['Name'] or 'conv_wgrad_kernel' in r['Name']:
        print("  %-70s calls/step %5.1f ms/step %.3f avg us %.1f"%(r['Name'][:70], int(r['Calls'])/9, float(r['TotalDurationNs'])/9/1e6, float(r['AverageNs'])/1e3))
PY
done
