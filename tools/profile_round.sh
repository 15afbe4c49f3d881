#!/bin/bash
# Collects this round's evidence on the GPU box into gpurun_out/<tag>_*.
# usage: tools/profile_round.sh r01
set -o pipefail
tag=${1:-r01}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --no-cpu-baseline --no-rehearsal > $out/${tag}_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 bench.py --no-cpu-baseline --no-rehearsal --no-parity --steps 1 --warmup 1 > $out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 bench.py --no-cpu-baseline --no-rehearsal --no-parity --steps 1 --warmup 1 > $out/${tag}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_valu -- python3 bench.py --no-cpu-baseline --no-rehearsal --no-parity --steps 1 --warmup 1 > $out/${tag}_valu.log 2>&1 || exit 1
python tools/pmc_summary.py $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write $out/${tag}_traffic.json $out/${tag}_valu $out/${tag}_bench.json > /dev/null || exit 1
cp $(ls $out/${tag}_trace/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
python tools/hostpath.py > $out/${tag}_hostpath.json 2>&1
tail -1 $out/${tag}_bench.json
cat $out/${tag}_hostpath.json | tail -1
