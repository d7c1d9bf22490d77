#!/bin/bash
# ON THE GPU BOX: SQ counters of the update kernel alone (tools/gemm_ab, K = 8192 SYRK, 8 matrices), one pass per counter group.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_gemm}
V=${2:-0}
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o pmc --output-format csv -- tools/gemm_ab 8 0 $V > $OUT/g$i.log 2>&1
  python3 - "$OUT/g$i" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm44" not in r.get("Kernel_Name", ""): continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:36s} {tot[k] / max(cnt[k], 1):18.1f} per dispatch  ({cnt[k]} dispatches)")
PY
  find $OUT/g$i -name "*.csv" -size +20M -delete
done
cat $OUT/summary.txt
