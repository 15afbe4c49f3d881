#!/bin/bash
# SQ counters of one program's kernels (one rocprofv3 pass per counter set;
# --pmc passes carry --kernel-trace only).
# usage: tools/pmc_kernel.sh <tag> <kernel-name substring> <run_program.py args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; pat=$2; shift 2
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
j=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES_EQ_64"; do
  j=$((j+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$j -- python3 tools/run_program.py "$@" > $out/s$j.log 2>&1
  tail -1 $out/s$j.log | cut -c1-160
done
python3 - <<PY
import csv, glob, collections, json, sys
agg=collections.defaultdict(float); n=collections.Counter(); names=set()
for f in glob.glob('$out/s*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$pat' in r['Kernel_Name']:
            names.add(r['Kernel_Name'])
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
res = {k: round(v / max(1, n[k]), 1) for k, v in sorted(agg.items())}
# (round 4 summarised a run of ANOTHER variant of the kernel under the benched
# one's label: the names the counters were taken on are part of the record, and
# exactly one kernel may match)
res['kernels'] = sorted(names)
res['launches_averaged'] = max(n.values()) if n else 0
print(json.dumps(res))
json.dump(res, open('$out/summary.json', 'w'), indent=1)
if len(names) != 1:
    sys.exit('pmc_kernel.sh: "$pat" matched %d kernels: %s' % (len(names), sorted(names)))
PY
