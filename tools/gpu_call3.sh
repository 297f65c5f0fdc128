set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in 0 1 2; do
TAPIR_CV_FORM=$f TAPIR_HIP_LIB=tools/bin/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what cvfusedtrace > gpurun_out/r04_cv_rows_trace_form$f.txt 2>&1; cat gpurun_out/r04_cv_rows_trace_form$f.txt | tail -8
TAPIR_CV_FORM=$f timeout 300 python tools/kbench.py --what cv --reps 20 --out gpurun_out/r04_kbench_cv_form$f.json > gpurun_out/r04_kbench_cv_form$f.log 2>&1; grep '"cost_volume_stage_fused"' gpurun_out/r04_kbench_cv_form$f.log
done
