#!/bin/bash
# SQ counter passes of one command (run ON THE GPU BOX): bash tools/pmc_sq.sh <kernel-name substring> <cmd...>
# prints per-launch means of the SQ activity / wait / instruction-mix counters of the matching kernels.
SUB=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_MFMA" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcsq_$i -o p -- "$@" > /dev/null 2> $R/gpurun_out/pmcsq_$i.err
done
cd $R
python3 - "$SUB" <<'PY'
import csv, glob, collections, sys
sub = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmcsq_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if sub not in k: continue
            acc[k[:60]][r['Counter_Name']] += float(r['Counter_Value']); disp[k[:60]].add(r['Dispatch_Id'])
        for k, v in acc.items():
            n = len(disp[k])
            print(k, 'launches', n, {a: round(b / n) for a, b in v.items()})
PY
rm -rf gpurun_out/pmcsq_*
