set -u
bash scripts/pmc_run.sh pmc_aecv "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" python3 scripts/pmc_probe.py aec
bash scripts/pmc_run.sh pmc_aecl "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python3 scripts/pmc_probe.py aec
bash scripts/pmc_run.sh pmc_aecw "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" python3 scripts/pmc_probe.py aec
