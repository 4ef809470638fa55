#!/bin/bash
# counters of the solve sweeps: what bounds the backward sweep?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bwd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^Counter_Name|Name *:" | grep -E "SQ_|TCP_|TA_|TCC_" | sed 's/.*:\s*//' | sort -u | tr '\n' ' ' > $OUT/avail.txt
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1"
run() {  # tag, counters...
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- $BENCH > /dev/null 2> $OUT/$tag.err
}
for deep in 0 4; do
  export DDH_BWD_DEEP=$deep
  run a$deep SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
  run b$deep SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS
  run c$deep SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_IFETCH
  run d$deep TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2
  run e$deep TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
  run f$deep TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_TAG_STALL_sum
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv, glob, collections, os
for tag in sorted(os.listdir("$OUT")):
    if not os.path.isdir(os.path.join("$OUT", tag)): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ddh::", "")[:44]
            if "solve_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(tag, k, "disp=%d" % len(n[k]), {c: "%.4g" % (x / len(n[k])) for c, x in v.items()})
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
