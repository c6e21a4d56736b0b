#!/bin/bash
# PMC passes over the bench to see where the GEMM launches wait (tuning).  bash tools/pmc_gemm.sh
# (a pass with TCP_* / TA_* counters hung the profiler on this pool and was dropped)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph > /dev/null 2> $OUT/p$i.log || echo "pass $i failed"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "conv2" if "persistent_kernel<2, 2, 1, 1, 0, 2>" in k else "conv1" if "persistent_kernel<2, 2, 1, 1, 0, 1>" in k else "head0(ring)" if "ring_kernel<3" in k else None
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print("==", name)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:]
        print(f"  {c:34s} {sum(v) / len(v):16.0f}")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
