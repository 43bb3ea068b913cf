source scripts/gpu_steps.sh
for abl in 0 1 2 3; do
  PA_PS_ABL=$abl step 200 gpurun_out/r6c_abl$abl.txt python scripts/conv_layers_ab.py --rounds 3 --only "det"
done
for abl in 0 1 2 3; do echo "== PA_PS_ABL=$abl"; cut -c1-100 gpurun_out/r6c_abl$abl.txt | grep "512->512\|256->512\|1024\|64->128\| 64-> 64\|sum"; done
