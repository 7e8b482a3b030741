#!/bin/bash
# PMC passes over tools/w4_pmc_case.py (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
# L2 hit / miss / fabric reads, L1 requests, wave wait / issue split, LDS.  Output: gpurun_out/w4pmc/<set>/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/w4pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TA_[A-Z0-9_]+|SQ_[A-Z0-9_]+|GRBM_[A-Z_]+)\b" | sort -u > $OUT/counters_available.txt
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -- python3 $R/tools/w4_pmc_case.py > $OUT/set$i.log 2>&1
  echo "set$i: $SET rc=$?" >> $OUT/sets.txt
done
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/w4pmc")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"conv_wino4_kernel<(\d), (\d)>", r["Kernel_Name"])
        if not m:
            continue
        a = acc[f"<{m.group(1)},{m.group(2)}>"][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(acc):
        fo.write(k + "\n")
        for c in sorted(acc[k]):
            v, n = acc[k][c]
            fo.write(f"   {c:34s} {v / n:16.1f}  (avg of {n} launches)\n")
print(open(out + "/summary.txt").read())
PY
