source scripts/gpu_steps.sh
L="512->512|256->512|1024|64->128| 64-> 64|128->128|sum"
step 200 gpurun_out/r6g_abl0.txt python scripts/conv_layers_ab.py --rounds 3 --only "det"
for n in 1 2 3; do
  PA_LIB_PATH=$PWD/build/libplayaid_psabl$n.so step 200 gpurun_out/r6g_abl$n.txt python scripts/conv_layers_ab.py --rounds 3 --only "det"
done
for n in 0 1 2 3; do echo "== PA_PS_ABL=$n"; cut -c1-100 gpurun_out/r6g_abl$n.txt | grep -E "$L"; done
bash scripts/pmc_psgemm.sh gpurun_out/r6_psgemm_pmc2 "256->512" && grep -A8 "psgemm" gpurun_out/r6_psgemm_pmc2/summary.txt
