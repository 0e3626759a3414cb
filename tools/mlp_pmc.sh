#!/bin/bash
set -e
out=gpurun_out/${1:-mlppmc}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/mlp_probe.py 20
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -o p -- python3 tools/mlp_probe.py 5 > $out/p$i.log 2>&1
  echo "pass $i done"
done
