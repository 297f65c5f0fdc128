set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out
timeout 900 python -m pytest tests/test_cycle_consistency.py tests/test_gpu_guard_bands.py tests/test_gpu_distributed.py -m gpu -x -q -s > $OUT/r04_t7.log 2>&1; grep -E "guarded tensors|passed|failed|Error|error" $OUT/r04_t7.log | cut -c1-300 | tail -12
for f in 1 2; do TAPIR_CV_FORM=$f timeout 300 python tools/kbench.py --what contraction --reps 20 --out $OUT/r04_kbench_contraction_form$f.json 2>&1 | grep contraction_ | cut -c1-420; done
KBENCH_MIXER_SHAPES=256x48 TAPIR_HIP_LIB=tools/bin/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what mixer --reps 20 --out $OUT/r04_kbench_mixer_fp8w.json 2>&1 | grep '"kernel"' | cut -c1-400
timeout 600 python tools/run_config5.py > $OUT/r04_config5_1gpu.json 2> $OUT/r04_config5.err; cat $OUT/r04_config5_1gpu.json
