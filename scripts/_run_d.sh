bash scripts/pmc_psgemm.sh gpurun_out/r6_psgemm_pmc "256->512" && cat gpurun_out/r6_psgemm_pmc/summary.txt
