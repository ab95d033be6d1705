#!/bin/bash
# tools/pmc_cmp.sh OUTDIR -- SQ/TCC counters of the plain A x SpMV for the block-slab (PDLP_MI355X_SLAB=1) and the
# wave-slab (=2) kernels, one rocprofv3 --pmc pass per counter group and layout (development tool).
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mkdir -p "$1" && cd "$1" && pwd)
for L in ${LAYOUTS:-1 2}; do
  export PDLP_MI355X_SLAB=$L
  KERNELS=${KERNELS:-spmv_ax_plain} bash $R/tools/pmc.sh $OUT/slab$L \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
    "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
    "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
    "TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_BUSY_avr GRBM_GUI_ACTIVE" \
    "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU" > $OUT/slab$L.txt 2>&1
done
