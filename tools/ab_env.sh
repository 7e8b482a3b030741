#!/bin/bash
# same-box A/B of bench.py under different environments:  tools/ab_env.sh "VAR=a" "VAR=b" ...   (GPU box)
# prints value / ms per step / dominant kernel frac and the SNGAN-64 leg of each
for e in "$@"; do
  env $e python bench.py --steps 20 --warmup 5 --no_cpu_baseline > /tmp/ab.json 2>/dev/null
  python - "$e" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
s = d.get("sngan64_conv_blocks", {})
print(f"{sys.argv[1]:32s} sngan32 {d['value']:8.1f} img/s {d['ms_per_step']:7.3f} ms  {d['roofline']['kernel']} frac {d['roofline']['frac']:.3f} | "
      f"sngan64 {s.get('images_per_s')} img/s frac_executed {s.get('frac_executed')}")
PY
done
