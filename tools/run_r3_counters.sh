# Round-3 counter evidence (GPU box, through gpurun): SQ counters per kernel for the GEMM family, cbhg_tail, bigru_scan
# and one dec_persistent launch; three separate --pmc passes (at most 8 SQ counters a pass), program directly after `--`.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES"
Bc="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"
Cc="GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for set in "$A" "$Bc" "$Cc"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_sq$i -- python3 tools/pmc_targets.py > $O/pmc_sq$i.log 2>&1 || { tail -5 $O/pmc_sq$i.log; exit 1; }
  f=$(find $O/pmc_sq$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_any.py $f > $O/sq_counters_$i.txt
  echo "pass $i done: $(wc -l < $O/sq_counters_$i.txt) lines"
  rm -rf $O/pmc_sq$i
done
