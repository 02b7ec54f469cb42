#!/bin/bash
# SQ counters of the GEMM kernels over a few shapes (development): bash tools/duo_pmc.sh [dbg] -> gpurun_out/duopmc<dbg>.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
DBG=${1:-0}
OUT=$R/gpurun_out/duopmc$DBG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OCRVI_DUO_DBG=$DBG
SH="61440,1536,384 122880,256,1024 61440,384,1536"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py f16x2 $SH > $OUT/a.log 2>&1 || echo "pass a failed"
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py f16x2 $SH > $OUT/b.log 2>&1 || echo "pass b failed"
cd $R
python3 - <<PY
import csv, glob, collections
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(out+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm_duo" not in k and "gemm_ring" not in k: continue
        key=(k[:60], r.get("Grid_Size",""), r.get("LDS_Block_Size",""))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key].add((f,r["Dispatch_Id"]))
for k,v in agg.items():
    wc=v.get("SQ_WAVE_CYCLES",1)/2  # counted in both passes
    gui=v.get("GRBM_GUI_ACTIVE",0)
    print(k)
    print("   mfma_busy %.3f" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(gui/8*1024) if gui else -1), " per wave-cycle: wait_any %.3f wait_inst %.3f valu %.3f lds %.3f sca %.3f wait_lds %.3f active_any %.3f" % tuple(v.get(c,0)/wc for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_ANY")), " insts valu %.0f salu %.0f lds %.0f mfma %.0f per dispatch-pass" % tuple(v.get(c,0)/max(len(n[k])/2,1) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_MFMA")))
PY
find $OUT -name "*.csv" -size +5M -delete
