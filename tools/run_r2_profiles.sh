# Round-2 profile set (run on the GPU box through gpurun): writes under gpurun_out/r02/, summaries are copied to profiles/r02/
set -o pipefail
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_r2.json 2> $O/bench_r2.err || exit 1
echo "bench done"; tail -c 400 $O/bench_r2.json
export GPU_MAX_HW_QUEUES=8
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mel -- python3 bench.py --no-cpu-baseline --no-extras > $O/stats_mel.log 2>&1 || exit 1
echo "stats mel done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wav -- python3 bench.py --to-wav --steps 60 --no-cpu-baseline --no-extras > $O/stats_wav.log 2>&1 || exit 1
echo "stats wav done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --to-wav --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --to-wav --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/pmc_write.log 2>&1 || exit 1
echo "pmc write done"
for d in pmc_fetch pmc_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $f > $O/$d.txt; head -8 $O/$d.txt; done
for d in stats_mel stats_wav; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; head -6 $f; done
unset GPU_MAX_HW_QUEUES
TACO_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --no-extras --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err; echo "gloo rc=$?"; tail -c 600 $O/gloo2.json
