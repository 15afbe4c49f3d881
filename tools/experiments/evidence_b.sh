#!/bin/bash
# Evidence of a round, part B: the other BASELINE configs (durations, PMC
# traffic), every config on one GPU, the reference corpus, the library's time
# model against the clock -> gpurun_out/<tag>_*.   usage: tools/experiments/evidence_b.sh r02
set -o pipefail
tag=${1:-r02}
out=gpurun_out
bash tools/profile_configs.sh $tag || exit 1
python tools/configs.py $out/${tag}_configs.json > $out/${tag}_configs.log 2> $out/${tag}_configs.err || exit 1
python tools/corpus_bench.py --reps 7 > $out/${tag}_corpus.jsonl 2> $out/${tag}_corpus.err || exit 1
python tools/corpus_bench.py --extent2 16384 16384 --only blur.soda sobel2d.soda --reps 9 >> $out/${tag}_corpus.jsonl 2>> $out/${tag}_corpus.err || exit 1
rm -f $out/${tag}_model_check.jsonl
python tools/model_check.py --out $out/${tag}_model_check.jsonl > /dev/null 2> $out/${tag}_model_check.err || exit 1
python tools/model_check.py --soda tests/golden/soda/heat3d.soda --extents 512x512x512 512x512x80 --fuse 2 --iterate 50 --out $out/${tag}_model_check.jsonl > /dev/null 2>> $out/${tag}_model_check.err || exit 1
python tools/copybench.py 1 > $out/${tag}_copybench.jsonl 2> /dev/null
python tools/copybench.py 2 >> $out/${tag}_copybench.jsonl 2> /dev/null
wc -l $out/${tag}_corpus.jsonl $out/${tag}_model_check.jsonl $out/${tag}_copybench.jsonl
