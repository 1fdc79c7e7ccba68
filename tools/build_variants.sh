#!/bin/bash
# builds tuning variants of libqsparse_hip.so into build_variants/ (development tool); every variant goes through
# __graft_entry__.build_hip (the translation units in parallel, objects cached under build/hip/<variant>)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
for U in 1 2 4; do for NT in 0 1; do for R in 8 16; do
  if [ "$R" = 16 ] && [ "$U$NT" != "21" ]; then continue; fi
  out=build_variants/libqs_u${U}_nt${NT}_r${R}.so
  python3 -c "import __graft_entry__ as g; g.build_hip(extra_defines=('QS_EW_UNROLL=$U','QS_EW_NT=$NT','QS_MEAN_ROWS_IN_FLIGHT=$R'), out='$out')"
done; done; done
ls -la build_variants
