#!/bin/bash
# Collects a round's evidence on the GPU box into gpurun_out/<tag>_*.  Run it
# as the LAST GPU act of a round, after the final edit to anything that enters
# a kernel's code (round 4 lost its counter evidence to a later soda_rt.h
# edit; tests/test_codegen.py::test_counter_evidence_is_for_the_kernels_head_builds
# now fails on the CPU when profiles/traffic.json names other machine code).
# usage: tools/profile_round.sh r05
set -o pipefail
tag=${1:-r01}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
quick="--no-cpu-baseline --no-rehearsal --no-other-configs --no-parity --steps 1 --warmup 1"
# 1. per-kernel time: rocprofv3 --kernel-trace --stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --no-cpu-baseline --no-rehearsal > $out/${tag}_trace.log 2>&1 || exit 1
# 2. counters, one pass per set (never together with --stats)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 bench.py $quick > $out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 bench.py $quick > $out/${tag}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_valu -- python3 bench.py $quick > $out/${tag}_valu.log 2>&1 || exit 1
# 3. the table bench.py reads, keyed by each kernel's machine code (the keys
#    come from the bench line of run 1)
python tools/pmc_summary.py $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write $out/${tag}_traffic.json $out/${tag}_valu $out/${tag}_trace.log > /dev/null || exit 1
cp $(ls $out/${tag}_trace/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
cp $out/${tag}_traffic.json profiles/traffic.json
# 4. the bench line itself, LAST: it must find its counters in the table
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver_like.json 2>> $out/${tag}_bench.err || exit 1
python tools/hostpath.py > $out/${tag}_hostpath.jsonl 2> $out/${tag}_hostpath.err
python - <<PY
import json
b = json.loads([l for l in open('$out/${tag}_bench.json') if l.startswith('{')][-1])
r = b['roofline']
print('value %.4g  ms/step %.4f  frac %.3f  bound %s  traffic %s  valu %s' % (
    b['value'], b['ms_per_step'], r['frac'], r['bound'], r.get('traffic'),
    (r.get('valu') or {}).get('frac_of_valu_issue_peak')))
assert r.get('traffic'), r.get('traffic_dropped')
PY
cat $out/${tag}_hostpath.jsonl
# 5. the enumeration behind codegen/hip/exact.py (1.9e9 operands, ~1 s): its
#    record is keyed by the text of the sequence the kernels use
#    (tests/test_exact.py::test_the_committed_enumeration_is_for_this_text)
python -m pytest tests/test_exact.py -m gpu -q -k every_operand > $out/${tag}_rsqrt_exact.log 2>&1 && \
  cp $out/rsqrt_exact.json $out/${tag}_rsqrt_exact.json
tail -1 $out/${tag}_rsqrt_exact.log
echo "(copy $out/${tag}_rsqrt_exact.json to profiles/rsqrt_exact.json if exact.py's text changed)"
