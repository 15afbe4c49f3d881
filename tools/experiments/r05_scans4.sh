#!/bin/bash
# Last scans of round 5, on the final code (after the wire-format and host-entry
# work): fresh seed ranges.  -> gpurun_out/r05_scans4.txt
out=gpurun_out/r05_scans4.txt
mkdir -p gpurun_out
: > $out
for spec in "generic 2300 2700" "rich 1200 1420" "window 760 860" "wire 1900 2400" "group 800 850" "ranks 340 360"; do
  echo "## python tools/fuzz_scan.py $spec" >> $out
  timeout -k 10 400 python tools/fuzz_scan.py $spec >> $out 2>&1
done
grep "seeds" $out
