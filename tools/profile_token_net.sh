#!/bin/bash
# usage (GPU box, repo root): tools/profile_token_net.sh <tag> [batch] [act]  -> gpurun_out/profiles/<tag>_token_net_kernel_stats.csv
# rocprofv3 kernel trace of the converted TokenNet's training step (tools/bench_token_net.py, QS_ONLY=pq)
tag=$1; batch=${2:-128}; act=${3:-gelu}
root=$(pwd)
out=$root/gpurun_out/profiles; mkdir -p $out
d=/tmp/prof_$tag; rm -rf $d
cd /tmp && export TMPDIR=/tmp QS_ONLY=pq
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o tok -- python3 $root/tools/bench_token_net.py $batch $act > $d.stdout 2> $d.stderr || { tail -20 $d.stderr; exit 1; }
stats=$(find $d -name '*kernel_stats.csv' | head -1)
cp $stats $out/${tag}_token_net_kernel_stats.csv
tail -1 $d.stdout
head -25 $stats | cut -c1-200
