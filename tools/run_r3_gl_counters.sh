# Round-3 SQ counters of the vocoder's iteration kernel (final state: register-plane exchanges): two --pmc passes over
# `tools/gl_bench.py 32 6`, program directly after `--`; summaries under gpurun_out/r03/.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
A="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
Bc="SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
for p in a b; do
  if [ $p = a ]; then set="$A"; else set="$Bc"; fi
  rm -rf $O/pmc_gl_$p
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $O/pmc_gl_$p -- python3 tools/gl_bench.py 32 6 > $O/pmc_gl_$p.log 2>&1 || { tail -5 $O/pmc_gl_$p.log; exit 1; }
  f=$(find $O/pmc_gl_$p -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_any.py $f > $O/gl_iter2_sq_counters_$p.txt
  grep -A9 "gl_iter2_kernel<1200, 300, true, true>" $O/gl_iter2_sq_counters_$p.txt | head -10
  rm -rf $O/pmc_gl_$p
done
