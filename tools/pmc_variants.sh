#!/bin/bash
# SQ counters of the fused jacobi2d kernel in three variants (one rocprofv3 pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcv
rm -rf $out; mkdir -p $out
i=0
for v in "--pipe 1" "--pipe 4 --chunk 200" "--pipe 4 --chunk 200 --shift bperm"; do
  i=$((i+1))
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/v${i}_$tag -- python3 tools/run_program.py jacobi2d.soda 8192 8192 --iterate 48 --fuse 12 $v --reps 2 > $out/v${i}_$tag.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3):
    agg=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob('gpurun_out/pmcv/v%d_*/*/*counter_collection.csv' % i):
        for r in csv.DictReader(open(f)):
            if 'T12' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    # per launch: rows are per dispatch (aggregated over dims?) -> average per dispatch
    disp = collections.Counter()
    print('variant', i, {k: round(v / max(1, n[k]), 1) for k, v in sorted(agg.items())})
PY
