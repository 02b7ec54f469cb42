set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ldspmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITERS=2
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/lds -- python3 $R/tools/ring_shapes.py f16x2 > $OUT/lds.log 2>&1 || echo "lds pass failed"
cd $R
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/ldspmc"
f=glob.glob(out+"/lds/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "gemm_ring" in k: print(k, {a:int(b) for a,b in v.items()})
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete
