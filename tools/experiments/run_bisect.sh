#!/bin/bash
# judges pre-compiled opt-bisect variants of fuzz613 (compiled on the CPU box into the JIT cache)
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  r=$(SODA_HIP_EXTRA_FLAGS="-mllvm -opt-bisect-limit=$L" timeout -k 10 100 python tools/experiments/triage_peel.py 613 '(1100, 207)' "dict(fuse=(2,), peel=-1)" 2>/dev/null | grep -o "('out0', [0-9]*" )
  echo "limit $L -> $r"
done
