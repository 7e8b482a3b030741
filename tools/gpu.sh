#!/bin/bash
# build the HIP library (cross-compile here), then run a command on the MI355X box
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C self-diagnosing-gan_amd/csrc
make -s -C oracle
T=${GPU_TIMEOUT:-900}
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
