#!/bin/bash
# Evidence of a round, part A: the bench line, its rocprofv3 kernel stats and
# PMC passes, the PCIe-inclusive rate, and one rank's slab of a 2/4/8-GPU run
# rehearsed on this GPU -> gpurun_out/<tag>_*.   usage: tools/experiments/evidence_a.sh r02
set -o pipefail
tag=${1:-r02}
bash tools/profile_round.sh $tag || exit 1
for n in 2 4 8; do
  python bench.py --emulate-slab $n --no-cpu-baseline > gpurun_out/${tag}_bench_slab$n.json 2> gpurun_out/${tag}_bench_slab$n.err || exit 1
  tail -1 gpurun_out/${tag}_bench_slab$n.json | cut -c1-200
done
