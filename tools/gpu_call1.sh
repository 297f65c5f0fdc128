set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_stages.py tests/test_tapnet_reference_pin.py tests/test_gpu_parity_full.py -m gpu -x -q -k "cost or cv or tapnet or head or full or golden or end_to_end" > gpurun_out/r04_t1.log 2>&1; tail -5 gpurun_out/r04_t1.log
timeout 300 python tools/kbench.py --what cv --reps 20 --out gpurun_out/r04_kbench_cv.json > gpurun_out/r04_kbench_cv.log 2>&1; cat gpurun_out/r04_kbench_cv.log | tail -8
timeout 600 python bench.py > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err; cat gpurun_out/r04_bench1.json | cut -c1-1500
