#!/bin/bash
# The bench lines of the other configurations and shard sizes, written under gpurun_out/<tag>/ (run on the GPU box from the
# repo root):  tools/bench_lines.sh <tag>;  copy the ones to keep into profiles/<tag>_bench_line_<name>.json
set -e
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" > $out/bench_line_$name.json 2> $out/bench_$name.err; echo "$name done"; }
run cartpole --config cartpole
run reach --config reach
run reach_xarm6 --config reach --robot xarm6
run h1 --config h1
run shadow --config shadow
run envs65536 --envs 65536
run envs262144 --envs 262144 --steps 50
run uniform --root-scatter uniform
