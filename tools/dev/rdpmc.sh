#!/bin/bash
# developer aid: PMC counters of the decode kernels for one answer type (separate passes, no tracing)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/rdpmc; rm -rf /tmp/rdpmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  DEC_ONLY=${1:-0x85} timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/rdpmc/p$i -o p -- python tools/dev/decbench.py 4096 > gpurun_out/rdpmc/run$i.log 2>&1
  f=$(find /tmp/rdpmc/p$i -name "*counter_collection.csv" | head -1); cp "$f" gpurun_out/rdpmc/pmc$i.csv
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/rdpmc/pmc*.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:28]
        if 'k_decode' in k or 'k_assemble' in k or 'k_segment' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
