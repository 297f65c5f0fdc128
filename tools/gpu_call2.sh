set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAPIR_HIP_LIB=tools/bin/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what cvfusedtrace > gpurun_out/r04_cv_rows_trace.txt 2>&1; cat gpurun_out/r04_cv_rows_trace.txt | tail -12
TAPIR_CV_QPW=8 TAPIR_HIP_LIB=tools/bin/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what cvfusedtrace > gpurun_out/r04_cv_rows_trace_q8.txt 2>&1; cat gpurun_out/r04_cv_rows_trace_q8.txt | tail -12
TAPIR_CV_QPW=8 timeout 300 python tools/kbench.py --what cv --reps 20 --out gpurun_out/r04_kbench_cv_q8.json > gpurun_out/r04_kbench_cv_q8.log 2>&1; grep '"cost_volume_stage_fused"' gpurun_out/r04_kbench_cv_q8.log
