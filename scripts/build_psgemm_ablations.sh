#!/bin/bash
# Diagnostic builds of the library with psgemm.hip's timing ablations compiled in (results wrong, timing only):
#   build/libplayaid_psabl1.so  no LDS-DMA copies after the prologue      build/libplayaid_psabl2.so  no matrix instructions
#   build/libplayaid_psabl3.so  both
# selected with PA_LIB_PATH (playaid_core_amd/_lib.py). Run here (hipcc cross-compiles), the .so files travel to the GPU box.
set -e
cd "$(dirname "$0")/.."
python -m playaid_core_amd._build > /dev/null
mkdir -p build
C=playaid_core_amd/csrc
OBJS=$(ls $C/*.o | grep -v psgemm.o)
for n in ${ABLS:-1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DPA_PS_ABL=$n ${PA_PS_EXTRA:-} -c $C/psgemm.hip -o build/psgemm_abl$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libplayaid_psabl$n.so $OBJS build/psgemm_abl$n.o
  echo build/libplayaid_psabl$n.so
done
