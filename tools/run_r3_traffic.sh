# Round-3 HBM-side traffic passes (separate runs per counter, program directly after `--`, no child process): FETCH_SIZE,
# WRITE_SIZE of the to-waveform command with one batch in flight; summaries under gpurun_out/r03/.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_$c; rm -rf $d
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --to-wav --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
  f=$(find $d -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $f > $O/bench_r3_to_wav_pmc_$c.txt; head -12 $O/bench_r3_to_wav_pmc_$c.txt; rm -rf $d
done
