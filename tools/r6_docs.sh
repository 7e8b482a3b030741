#!/bin/bash
# profiles/r06_* from the output of tools/r6_final.sh (gpurun_out/prof_r06, gpurun_out/r6raw): kernel-stat summaries, PMC traffic
# (+ profiles/pmc_traffic.json), MFMA busy, raw bench lines.   usage (repo root): bash tools/r6_docs.sh
set -e
P=gpurun_out/prof_r06
python tools/round_docs.py r06 $P gpurun_out/r6raw/bench.json > /dev/null
f32=$(ls $P/pmc32_FETCH_SIZE/*/*counter_collection.csv | head -1); w32=$(ls $P/pmc32_WRITE_SIZE/*/*counter_collection.csv | head -1)
f64=$(ls $P/pmc64_FETCH_SIZE/*/*counter_collection.csv | head -1); w64=$(ls $P/pmc64_WRITE_SIZE/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $f32 $w32 sngan32 /tmp/_t32.md > /dev/null
python tools/pmc_traffic.py $f64 $w64 sngan64 /tmp/_t64.md > /dev/null
{
  echo "# r06: HBM traffic per kernel from the PMC counters (final round-6 build)"
  echo
  echo "Collected by \`tools/profile_round.sh r06\` (separate \`rocprofv3 --kernel-trace --pmc FETCH_SIZE\` / \`WRITE_SIZE\` passes of \`python3 bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timer\`); bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB as \`MI355X_MICROARCH.md\` prescribes for gfx950.  \`pmc_traffic.json\` (what \`bench.py\` quotes as \`roofline.traffic\`) is regenerated from these passes (\`tools/r6_docs.sh\`)."
  echo
  cat /tmp/_t32.md
  echo
  cat /tmp/_t64.md
} > profiles/r06_pmc_traffic.md
m32=$(ls $P/pmc32_mfma/*/*counter_collection.csv | head -1); m64=$(ls $P/pmc64_mfma/*/*counter_collection.csv | head -1)
python tools/pmc_mfma.py "r06: MFMA-busy per kernel (final round-6 build)" profiles/r06_mfma_util.md "SNGAN-32 (default bench.py workload)=$m32" "SNGAN-64=$m64" > /dev/null
mkdir -p profiles/r06_raw
cp gpurun_out/r6raw/*.json profiles/r06_raw/
head -30 profiles/r06_sngan32_summary.md | cut -c1-160
cp gpurun_out/r6raw/*.txt profiles/r06_raw/
cp gpurun_out/prof_r06/sg2_summary.md profiles/r06_sg2_256_summary.md
cp $(ls gpurun_out/prof_r06/sg2/*/*kernel_stats.csv | head -1) profiles/r06_sg2_256_kernel_stats.csv
