#!/bin/bash
# build libtsl_hip.so (same command as __graft_entry__.build); prints errors only
set -o pipefail
cd "$(dirname "$0")/../thinshelllab_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics "$@" -o ../lib/libtsl_hip.so tsl_hip.hip 2>&1 | grep -E " error|fatal" -A4 | head -60
exit ${PIPESTATUS[0]}
