#!/bin/bash
# SQ counter passes over tools/pmc_driver.py.  usage: bash tools/gpu_sq.sh <tag> <B>
TAG=${1:-sq}
B=${2:-16}
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/${TAG}${i} -o $TAG -- python tools/pmc_driver.py $B > /dev/null 2> gpurun_out/${TAG}${i}.err
  tail -2 gpurun_out/${TAG}${i}.err
done
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/${TAG}?/*counter_collection.csv") + glob.glob("gpurun_out/${TAG}?/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("sdp_"):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
