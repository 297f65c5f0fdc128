cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/kbench.py --what cv --reps 30 --out gpurun_out/r04_kbench_cv_nospill.json 2>&1 | grep '"kernel"' | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_stages.py -m gpu -x -q -k "cost or cv or golden" 2>&1 | tail -2
