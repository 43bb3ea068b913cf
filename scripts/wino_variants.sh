#!/bin/bash
# Build alternative libraries that differ in wino.hip's compile-time knobs only (-> build/libplayaid_<tag>.so), for A/B runs on
# the GPU box through PA_LIB_PATH. Usage: scripts/wino_variants.sh "<tag>:<defines>" ...   e.g.  "a5:-DWN_AHEAD=5" "q:-DWN_A128=1"
set -eu
cd "$(dirname "$0")/.."
mkdir -p build
OBJS=$(ls playaid_core_amd/csrc/*.o | grep -v "/wino.o")
for v in "$@"; do
  tag=${v%%:*}; defs=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $defs -c playaid_core_amd/csrc/wino.hip -o build/wino_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libplayaid_$tag.so $OBJS build/wino_$tag.o
  echo "built build/libplayaid_$tag.so ($defs)"
done
