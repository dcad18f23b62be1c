#!/bin/bash
# GPU tool: rocprofv3 PMC passes (counters only, no trace domains) over one fp32-emulating GEMM shape (gemm_bf16x3).
# usage: pmc_gemm_f32x3.sh <tag> <layout M N K>; summary: python tools/pmc_summary.py <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_$tag/p$i -- python $GRAFT_REPO_ROOT/tools/gemm_f32x3_one.py "$@" 10 > $GRAFT_REPO_ROOT/gpurun_out/pmcb_$tag.p$i.log 2>&1
done
