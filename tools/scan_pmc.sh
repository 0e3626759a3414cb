#!/bin/bash
# Counter passes over tools/scan_probe.py; one rocprofv3 run per pass (block limits: 8 SQ, 4 TCP, 2 TA, 4 TCC).
#   tools/scan_pmc.sh <tag>   ->  gpurun_out/<tag>/p*/ ; summarise with tools/pmc_summary.py gpurun_out/<tag> lattice
set -e
out=gpurun_out/${1:-scanpmc}
mkdir -p $out
export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -o p -- python3 tools/scan_probe.py 3 > $out/p$i.log 2>&1
  echo "pass $i done"
done
