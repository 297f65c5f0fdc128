timeout 300 python tools/kbench.py --what mixer --dtypes bfloat16 --out gpurun_out/r02ap_kbench_mixer.json 2>&1 | grep pips_mixer > gpurun_out/r02ap_kbench_mixer.txt
python - <<'PY'
import json
for l in open('gpurun_out/r02ap_kbench_mixer.txt'):
    d = json.loads(l); print(d['kernel'][11:], d['N'], d['T'], d['med_us'], d.get('tflops'), d.get('max_abs_diff_vs_separate'))
PY
timeout 300 python -m pytest tests -m gpu -x -q -k "hot_path or config2 or f32_full" 2>&1 | tail -2
