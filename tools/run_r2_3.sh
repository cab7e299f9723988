set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t3.log; tail -3 gpurun_out/r2_t3.log
GPU_MAX_HW_QUEUES=4 timeout -k 10 200 python bench.py --keep-hw-queues --no-extras --no-roofline --no-cpu-baseline --steps 50 > gpurun_out/r2_child.json 2> gpurun_out/r2_child.err; echo child rc=$?; tail -c 300 gpurun_out/r2_child.json; tail -c 500 gpurun_out/r2_child.err
timeout -k 10 120 python tools/gl_bench.py 32 60 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_gl_a -- python3 tools/gl_bench.py 32 6 > gpurun_out/pmc_gl_a.log 2>&1; echo pmcA rc=$?
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc_gl_b -- python3 tools/gl_bench.py 32 6 > gpurun_out/pmc_gl_b.log 2>&1; echo pmcB rc=$?
for d in pmc_gl_a pmc_gl_b; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_any.py $f gl_iter > gpurun_out/$d.txt; tail -12 gpurun_out/$d.txt; done
