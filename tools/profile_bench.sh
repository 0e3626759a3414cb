#!/bin/bash
# Profiles of the bench command, written under gpurun_out/<tag>/ (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag>
# 1. bench line (with the CPU baseline)   2. kernel trace + stats of the same command   3./4. FETCH_SIZE and WRITE_SIZE
# counter passes   5. SQ counters (VALU / MFMA busy, instructions, waves).  The counter passes run the step launch by
# launch (--eager): the counters are collected per dispatch.
# tools/reduce_profiles.py <tag> then turns them into the committed summaries under profiles/.
set -e
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench.err
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/trace.log 2>&1
echo "trace done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
echo "write done"
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq -o p -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_sq.log 2>&1
echo "sq done"
