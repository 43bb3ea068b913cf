#!/bin/bash
# timing-only ablations of the igemm main loop (needs build/libplayaid_abl.so, -DPA_ABLATION_BUILD)
export PA_LIB_PATH=$PWD/build/libplayaid_abl.so
for tile in ${TILES:-2 1}; do
for abl in 0 1 4 5 8 9; do
  echo "== tile $tile ablate $abl (1 no global->LDS loads, 4 no barrier, 8 no mfma)"
  PA_FORCE_TILE=$tile PA_FORCE_SPLITK=1 PA_ABLATE=$abl python scripts/layer_times.py 2>&1 | grep -E "layer1.0.conv1|layer2.1.conv1|layer3.1.conv1"
done
done
