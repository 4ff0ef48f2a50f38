#!/bin/bash
# Round 3: SQ counters of the lane-per-pixel PatchMatch colour pass (k_mvs_propagate_lane, PVLM_MVS_LANE=1) next to the wave-per-pixel one
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3lane; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/tools/mvs_bench.py"
for L in ${LANES:-1 0}; do
  export PVLM_MVS_LANE=$L
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$L -- $W > $O/trace$L.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc$L -- $W > $O/pmc$L.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmcb$L -- $W > $O/pmcb$L.log 2>&1
  (cd $R && python tools/pmc_kernels.py $O/pmc_lane$L.json '{"k_mvs_propagate": 518400}' $O/trace$L $O/pmc$L k_mvs_propagate | tail -40)
  (cd $R && python - <<P
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmcb$L/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_mvs_propagate" in k: agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,c in agg.items(): print(k, dict(c))
P
)
done
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
