#!/bin/bash
# builds tuning variants of libqsparse_hip.so into build_variants/ (development tool)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
for U in 1 2 4; do for NT in 0 1; do for R in 8 16; do
  if [ "$R" = 16 ] && [ "$U$NT" != "21" ]; then continue; fi
  out=build_variants/libqs_u${U}_nt${NT}_r${R}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
     -DQS_EW_UNROLL=$U -DQS_EW_NT=$NT -DQS_MEAN_ROWS_IN_FLIGHT=$R qsparse_amd/csrc/qsparse_hip.hip -o $out &
done; done; done
wait
ls -la build_variants
