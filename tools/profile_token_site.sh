#!/bin/bash
# usage (GPU box, repo root): tools/profile_token_site.sh  -> gpurun_out/profiles/r06_token_site_kernel_stats.txt
# rocprofv3 --kernel-trace --stats of the composite token-major site (tools/bench_token_site.py, QS_NO_EVENTS=1) on two shapes
root=$(pwd); out=$root/gpurun_out/profiles; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
log=$out/r06_token_site_kernel_stats.txt; : > $log
for sh in 256,197,3072 64,1024,4096; do
  d=/tmp/tok_$sh; rm -rf $d
  QS_NO_EVENTS=1 QS_SHAPE=$sh rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $root/tools/bench_token_site.py > $d.txt 2>&1
  grep composite $d.txt >> $log
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 $root/tools/summarize_token_stats.py "$f" >> $log
done
cat $log
