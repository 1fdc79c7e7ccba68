# Fresh-seed randomised campaigns next to tools/run_campaigns.sh (GPU box, repo root): seeds no earlier round used -> gpurun_out/extra/*.txt
# (round 6: 18,000 cases, 0 failures; summaries in profiles/r06_extra_campaigns/)
mkdir -p gpurun_out/extra
run() { name=$1; shift; env "$@" 2>&1 | grep -E "^FAIL|^fuzz" | cut -c1-900 > gpurun_out/extra/$name.txt; tail -1 gpurun_out/extra/$name.txt; }
run cpu_gpu_101   QS_X=1 python3 tests/fuzz/fuzz_cpu_gpu.py 3000 101
run tok_303       QS_FUZZ_WHAT=tok python3 tests/fuzz/fuzz_cpu_gpu.py 3000 303
run image_404     QS_X=1 python3 tests/fuzz/fuzz_image.py 3000 404
run parity_505    QS_X=1 python3 tests/fuzz/fuzz_parity.py 2000 505
run graph_606     QS_FUZZ_GRAPH=1 python3 tests/fuzz/fuzz_cpu_gpu.py 2000 606
run exchange_707  QS_FUZZ_EXCHANGE=1 python3 tests/fuzz/fuzz_cpu_gpu.py 2000 707
run tok_graph_808 QS_FUZZ_GRAPH=1 QS_FUZZ_WHAT=tok python3 tests/fuzz/fuzz_cpu_gpu.py 1500 808
run tok_exch_909  QS_FUZZ_EXCHANGE=1 QS_FUZZ_WHAT=tok python3 tests/fuzz/fuzz_cpu_gpu.py 1500 909
