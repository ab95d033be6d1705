#!/bin/bash
# is the slab SpMV instruction-bound?  SQ counters of the plain A x kernel on the random 1M LP and on the structured LP
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03o
for cfg in rand struct; do
  if [ $cfg = struct ]; then export KBENCH_ARGS="--structured"; else unset KBENCH_ARGS; fi
  KERNELS=spmv_ax_plain tools/pmc.sh $O/$cfg "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
     "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
     "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
     "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > $O.$cfg.log 2>&1
  tail -40 $O.$cfg.log
done
