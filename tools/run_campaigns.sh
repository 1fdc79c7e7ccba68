#!/bin/bash
# usage (GPU box, repo root): tools/run_campaigns.sh <tag> [scale]   -> gpurun_out/profiles/<tag>_fuzz_*.txt (one-line summaries + failures)
# The randomised differential campaigns of tests/fuzz/ on the tree as it is (keep the summaries of the final
# kernels): CPU path vs HIP path of modules (NaN / Inf / -Inf in 30 % of the cases), of converted weight networks, of the functional
# API; sites vs the oracle's state machines; the same with the statistics exchange live on a one-rank RCCL group; with steady-state
# steps replayed from a hipGraph; the autocast image route against the plain route under torch.autocast.
tag=${1:-r06}; scale=${2:-1}
out=gpurun_out/profiles; mkdir -p $out
sha=$(cat .tree_sha 2>/dev/null || echo unknown)
run() {  # name, env..., -- command
    name=$1; shift
    log=$out/${tag}_fuzz_${name}.txt
    echo "# tree $sha; $*" > $log
    env "$@" 2>&1 | grep -E "^FAIL|^fuzz" | cut -c1-700 >> $log
    tail -1 $log
}
run cpu_gpu_4242      QS_X=1 python3 tests/fuzz/fuzz_cpu_gpu.py $((3000*scale)) 4242
run cpu_gpu_7         QS_X=1 python3 tests/fuzz/fuzz_cpu_gpu.py $((3000*scale)) 7
run token_major       QS_FUZZ_WHAT=tok python3 tests/fuzz/fuzz_cpu_gpu.py $((2000*scale)) 17
run nets              QS_FUZZ_WHAT=net python3 tests/fuzz/fuzz_cpu_gpu.py $((2000*scale)) 11
run functional        QS_FUZZ_MODE=functional python3 tests/fuzz/fuzz_cpu_gpu.py $((4000*scale)) 12
run oracle_sites      QS_X=1 python3 tests/fuzz/fuzz_parity.py $((2000*scale)) 13
run exchange          QS_FUZZ_EXCHANGE=1 python3 tests/fuzz/fuzz_cpu_gpu.py $((2000*scale)) 14
run graph             QS_FUZZ_GRAPH=1 python3 tests/fuzz/fuzz_cpu_gpu.py $((2000*scale)) 15
run image             QS_X=1 python3 tests/fuzz/fuzz_image.py $((3000*scale)) 16
