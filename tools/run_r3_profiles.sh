# Round-3 profile set (GPU box, through gpurun): rocprofv3 kernel stats of the default bench command and of the
# to-waveform command (program directly after `--`; --no-extras: no child process under the profiler), the 2-rank gloo
# rehearsal of the N=2 launch (self-launched by `bench.py --gpus 2`), outputs under gpurun_out/r03/.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mel -- python3 bench.py --no-cpu-baseline --no-extras > $O/stats_mel.log 2>&1 || { tail -5 $O/stats_mel.log; exit 1; }
echo "stats mel done"; grep -h '^{' $O/stats_mel.log | tail -c 600
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wav -- python3 bench.py --to-wav --steps 60 --no-cpu-baseline --no-extras > $O/stats_wav.log 2>&1 || { tail -5 $O/stats_wav.log; exit 1; }
echo "stats wav done"
for d in stats_mel stats_wav; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; head -6 $f; rm -rf $O/$d; done
unset GPU_MAX_HW_QUEUES
TACO_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-extras --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err; echo "gloo rc=$?"; tail -c 700 $O/gloo2.json
