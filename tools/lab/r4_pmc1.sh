set -e
export WMF_ITER_EPS=${WMF_ITER_EPS:-6e-8}
bash tools/prof.sh r4_pmc_a "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" cfg3 items 0 3 | grep -A6 "solve_iter"
bash tools/prof.sh r4_pmc_b "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" cfg3 items 0 3 | grep -A6 "solve_iter"
bash tools/prof.sh r4_pmc_c "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" cfg3 items 0 3 | grep -A6 "solve_iter"
