#!/bin/bash
# SQ counters of the fused jacobi2d T=12 kernel in several variants (one
# rocprofv3 pass per counter set; --pmc passes carry --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_r02
rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
i=0
for v in "--pipe 1 --shift dpp" "--pipe 1 --shift swzh" "--pipe 4 --shift dpp" "--pipe 4 --shift swzh" "--pipe 1 --shift none"; do
  i=$((i+1))
  j=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
             "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES_EQ_64"; do
    j=$((j+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/v${i}_s$j -- python3 tools/run_program.py jacobi2d.soda 8192 8192 --iterate 48 --fuse 12 $v --reps 2 > $out/v${i}_s$j.log 2>&1
    tail -1 $out/v${i}_s$j.log | cut -c1-200
  done
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for i in (1,2,3,4,5):
    agg=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob('gpurun_out/pmc_r02/v%d_s*/**/*counter_collection.csv' % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'T12' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    res['variant%d' % i] = {k: round(v / max(1, n[k]), 1) for k, v in sorted(agg.items())}
    print('variant', i, res['variant%d' % i])
json.dump(res, open('gpurun_out/pmc_r02/summary.json', 'w'), indent=1)
PY
