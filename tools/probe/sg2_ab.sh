#!/bin/bash
# same-box A/B of the StyleGAN2 256 x 256 iteration under different environments:  tools/probe/sg2_ab.sh "VAR=a" "VAR=b" ...  (GPU box)
# (boxes differ by up to ~10 % on matrix-pipe-bound work: compare within one call only)
STEPS=${SG2_STEPS:-16}
for e in "$@"; do
  env $e python bench.py --workload stylegan2 --steps $STEPS --warmup 3 > /tmp/ab.json 2>/dev/null
  python - "$e" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:44s} {d['value']:8.2f} img/s {d['ms_per_step']:8.3f} ms")
PY
done
