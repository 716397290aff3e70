set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/tools/ubench/conv_time.py 12 > $OUT/run1.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -o b -- python3 $ROOT/tools/ubench/conv_time.py 12 > $OUT/run2.txt 2>&1 || true
python3 - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"][:50]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])] += 1
    for k in acc:
        if "conv_deep" in k: print(d, k, {c: v / max(cnt[(k,c)],1) for c, v in acc[k].items()})
PY
tail -3 $OUT/run1.txt
rm -rf $OUT/p1 $OUT/p2
