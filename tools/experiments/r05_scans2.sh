#!/bin/bash
# A second, longer round of random scans on the final code of round 5.
out=gpurun_out/r05_scans2.txt
: > $out
run() {
  echo "## $*" >> $out
  "$@" >> $out 2>&1      # (straight into the file: a pipe into tail looks hung)
  tail -1 $out
}
run python tools/fuzz_scan.py generic 1600 2300
run python tools/fuzz_scan.py rich 760 1200
run python tools/fuzz_scan.py window 620 900
run python tools/fuzz_scan.py options 820 1100
run python tools/fuzz_scan.py group 700 900
run python tools/fuzz_scan.py deep 460 540
run python tools/fuzz_scan.py wide 680 900
run python tools/fuzz_scan.py ranks 300 380
