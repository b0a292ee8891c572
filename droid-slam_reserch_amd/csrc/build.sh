#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU).
#   usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../droid_backends/libdroid_backends_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
  "$here/api.hip" "$here/ba_kernels.hip" "$here/chol.hip" "$here/corr.hip" "$here/geom.hip" \
  -o "$out" "$@"
echo "built $out"
