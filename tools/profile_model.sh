#!/bin/bash
# usage (GPU box, repo root): tools/profile_model.sh <tag> <arch> <batch> [steps]   -> gpurun_out/profiles/<tag>_family_table.txt
# rocprofv3 kernel trace of tools/profile_config.py + the per-family table of the library's kernels (tools/family_table.py)
set -e
tag=$1; arch=${2:-resnet50}; batch=${3:-256}; steps=${4:-5}
root=$(pwd)
out=$root/gpurun_out/profiles; mkdir -p $out
d=/tmp/prof_$tag; rm -rf $d
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o cfg -- python3 $root/tools/profile_config.py $arch $batch $steps > $d.stdout 2> $d.stderr || { tail -20 $d.stderr; exit 1; }
meta=$(grep '^{' $d.stdout | tail -1)
trace=$(find $d -name '*kernel_trace.csv' | head -1)
stats=$(find $d -name '*kernel_stats.csv' | head -1)
python3 $root/tools/family_table.py $trace "$meta" > $out/${tag}_family_table.txt
cp $stats $out/${tag}_model_kernel_stats.csv
echo "$meta" > $out/${tag}_model_meta.json
cat $out/${tag}_family_table.txt | head -40
