#!/bin/bash
# A second, longer round of random scans on the final code of round 5, in two
# parts (one gpurun call may run 20 minutes).
part=${1:-a}
out=gpurun_out/r05_scans2_$part.txt
: > $out
run() {
  echo "## $*" >> $out
  "$@" >> $out 2>&1      # (straight into the file: a pipe into tail looks hung)
  tail -1 $out
}
if [ "$part" = a ]; then
  run python tools/fuzz_scan.py generic 1600 2300
  run python tools/fuzz_scan.py rich 760 1200
else
  run python tools/fuzz_scan.py window 620 760
  run python tools/fuzz_scan.py options 820 950
  run python tools/fuzz_scan.py group 700 800
  run python tools/fuzz_scan.py deep 460 500
  run python tools/fuzz_scan.py wide 680 800
  run python tools/fuzz_scan.py ranks 300 340
fi
