#!/bin/bash
# <app>_kernel on wire streams with the outputs stored in place by the program
# (stream.emit_late) against the shift + copy pass of rounds 1-4, device-
# resident banks.  -> gpurun_out/r05_wire_direct.jsonl
set -e
out=gpurun_out/r05_wire_direct.jsonl
mkdir -p gpurun_out
: > $out
g=tests/golden/soda
python tools/streambench.py --soda $g/blur.soda --extent 2000 16384 >> $out
python tools/streambench.py --soda $g/blur.soda --tile 16384 --extent 16384 16384 >> $out
python tools/streambench.py --soda $g/jacobi2d.soda --tile 8192 --iterate 1 --extent 8192 8192 >> $out
python tools/streambench.py --soda $g/jacobi2d.soda --tile 8192 --iterate 100 --extent 8192 8192 --steps 3 >> $out
python tools/streambench.py --soda $g/heat3d.soda --extent 32 32 65536 >> $out
python tools/streambench.py --soda $g/heat3d.soda --tile 512 512 --iterate 2 --extent 512 512 512 >> $out
python tools/streambench.py --soda $g/sobel2d.soda --tile 4096 --extent 4096 8192 >> $out
python tools/streambench.py --soda $g/blur.soda --tile 16384 --banks 4 --extent 16384 16384 >> $out
python tools/streambench.py --soda $g/jacobi2d.soda --tile 8192 --banks 2 --iterate 1 --extent 8192 8192 >> $out
cat $out
