#!/bin/bash
# Instruction-cache counters of the GEMM kernels over a few shapes (development): bash tools/ring_icache_pmc.sh -> gpurun_out/icache/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SH="61440,1536,384 122880,256,1024 61440,384,1536"
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py f16x2 $SH > $OUT/a.log 2>&1 || echo "pass a failed"
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py f16x2 $SH > $OUT/b.log 2>&1 || echo "pass b failed"
cd $R
python3 - <<PY
import csv, glob, collections
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(out+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm_duo" not in k and "gemm_ring" not in k: continue
        key=(k[:70], r.get("Grid_Size",""))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key].add((f,r["Dispatch_Id"]))
for k,v in agg.items():
    print(k, "dispatches", len(n[k]))
    print("   ", {c: round(x) for c,x in sorted(v.items())})
PY
find $OUT -name "*.csv" -size +5M -delete
