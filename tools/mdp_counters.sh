#!/bin/bash
# Instruction and wait counters of the fused MDP kernel alone (run on the GPU box from the repo root):
#   tools/mdp_counters.sh <tag> [config] [envs] [phases]     -> gpurun_out/<tag>_sq*/
set -e
tag=$1; cfg=${2:-anymal}; envs=${3:-16384}; phases=${4:-31}
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_sq1 -o p -- python3 tools/mdp_once.py $cfg $envs $phases > gpurun_out/${tag}_sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d gpurun_out/${tag}_sq2 -o p -- python3 tools/mdp_once.py $cfg $envs $phases > gpurun_out/${tag}_sq2.log 2>&1
