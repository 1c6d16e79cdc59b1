#!/bin/bash
# rocprofv3 SQ counters for the default bench workload (one --pmc pass, 8 SQ counters)
set -e
TAG=${1:-r01sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/sq_stdout.log 2>&1
python3 - <<PY
import csv, collections, json, sys
rows = list(csv.DictReader(open("$OUT/sq/bench_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "ns2d_fast" not in k: continue
    disp = len({r["Dispatch_Id"] for r in rows if r["Kernel_Name"].startswith(k[:30])})
    out = {"kernel": k, "dispatches": disp, **{c: v / max(disp, 1) for c, v in d.items()}}
    wc = out.get("SQ_WAVE_CYCLES", 0) or 1
    out["frac_of_wave_cycles"] = {c: round(out[c] / wc, 4) for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if c in out}
    print(json.dumps(out))
PY
