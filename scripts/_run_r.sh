source scripts/gpu_steps.sh
rm -f gpurun_out/r6r_stamps.txt
PA_LIB_PATH=$PWD/build/libplayaid_psstamp.so PA_PS_STAMP_FILE=$PWD/gpurun_out/r6r_stamps.txt step 300 gpurun_out/r6r_probe.txt python scripts/conv_layers_ab.py --rounds 1 --only "det"
sort gpurun_out/r6r_stamps.txt | uniq -c | sort -rn | awk '{$1="";print}' | sort -u | head -40
