#!/bin/bash
# The GPU minutes left at the very end of round 5, spent on fresh seeds.
# -> gpurun_out/r05_scans5.txt
out=gpurun_out/r05_scans5.txt
mkdir -p gpurun_out
: > $out
for spec in "deep 500 540" "options 1150 1330" "wire 3100 3350" "wide 1100 1250"; do
  echo "## python tools/fuzz_scan.py $spec" >> $out
  timeout -k 10 230 python tools/fuzz_scan.py $spec >> $out 2>&1
done
grep "seeds\|differ" $out
