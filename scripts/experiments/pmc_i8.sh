cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm1 /tmp/pm2
LBFGSX_GRAM=i8 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d /tmp/pm1 -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 14 > /dev/null 2>&1
LBFGSX_GRAM=i8 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_BUSY_CYCLES --output-format csv -d /tmp/pm2 -o b -- python $GRAFT_REPO_ROOT/scripts/bench_lbfgsb.py --n 1e7 --iters 14 > /dev/null 2>&1
python3 - <<PY
import csv, collections
for d in ("/tmp/pm1", "/tmp/pm2"):
    try:
        rows = list(csv.DictReader(open(d + "/b_counter_collection.csv")))
    except Exception as e:
        print(d, "ERR", e); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if "k_gram_i8<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-28s n=%d  max %.4g  mean %.4g" % (k, len(v), max(v), sum(v) / len(v)))
PY
