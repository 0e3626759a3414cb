#!/bin/bash
# Profiles of the bench command, written under gpurun_out/<tag>/ (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag>
# 1. bench line (with the CPU baseline)   2. kernel trace + stats   3./4. FETCH_SIZE and WRITE_SIZE counter passes.
# tools/reduce_profiles.py <tag> then turns them into the committed summaries under profiles/.
set -e
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench.err
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/trace.log 2>&1
echo "trace done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
echo "write done"
