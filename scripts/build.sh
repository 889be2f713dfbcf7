#!/bin/bash
# build libtsl_hip.so (same command as __graft_entry__.build); prints errors only
set -o pipefail
cd "$(dirname "$0")/../thinshelllab_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form "$@" -o ../lib/libtsl_hip.so tsl_hip.hip 2>&1 | grep -E " error|fatal" -A4 | head -60
rc=${PIPESTATUS[0]}
# record the digest of the sources the library was built from (what __graft_entry__.build() compares)
[ $rc -eq 0 ] && [ $# -eq 0 ] && python3 - <<PY
import sys; sys.path.insert(0, "../..")
import __graft_entry__ as g, os
srcs = [os.path.join(g.CSRC, f) for f in os.listdir(g.CSRC)] + [os.path.join(g.ROOT, "include", "tsl_hip.h")]
open(g.LIB + ".sources", "w").write(g._sources_digest(srcs) + " " + " ".join(g.HIPCC_FLAGS) + "\n")
PY
exit $rc
