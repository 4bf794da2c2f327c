#!/bin/bash
# Counter passes over tools/attn_pmc.py (separate --pmc runs, kernel trace only).  On the GPU box:
#   bash tools/attn_pmc.sh <tag>   -> gpurun_out/<tag>_pmc{1,2,3}/...counter_collection.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-attn}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
    rm -rf $R/gpurun_out/${tag}_pmc$i
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/${tag}_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/${tag}_pmc$i.log; }
    i=$((i+1))
done
python3 $R/tools/attn_pmc_parse.py $R/gpurun_out/${tag}_pmc1 $R/gpurun_out/${tag}_pmc2 $R/gpurun_out/${tag}_pmc3 > $R/gpurun_out/${tag}_pmc.json
cat $R/gpurun_out/${tag}_pmc.json
