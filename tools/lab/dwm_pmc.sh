#!/bin/bash
# On the GPU box: counters of the matrix-pipe depthwise kernel beside the column kernel, lab binary build/lab/dwm_lab (bf16, B = 64).
#   bash tools/lab/dwm_pmc.sh [waves]
R=$PWD; O=$R/gpurun_out/dwm_pmc; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
W=${1:-2048}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $R/build/lab/dwm_pmcbin $W > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/build/lab/dwm_pmcbin $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/build/lab/dwm_pmcbin $W > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/tcc -- $R/build/lab/dwm_pmcbin $W > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/inst -- $R/build/lab/dwm_pmcbin $W > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/dwm_pmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dwconv7" not in k: continue
        name = k.split("(")[0].replace("void acx::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(agg):
    print(name)
    for c, v in sorted(agg[name].items()):
        print("   %-28s %14.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
