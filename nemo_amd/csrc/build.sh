#!/bin/bash
# Builds libnemo_hip.so for gfx950 (MI355X).  -ffp-contract=off is part of the
# numerical contract: coverage must not depend on FMA contraction.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$ROOT/nemo_amd/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Wno-unused-value \
  -I"$ROOT/include" -o "$ROOT/nemo_amd/lib/libnemo_hip.so" "$HERE/nemo_hip.hip" "$@"
echo "built $ROOT/nemo_amd/lib/libnemo_hip.so"
