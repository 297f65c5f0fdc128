mkdir -p gpurun_out/r02n
timeout 250 python tools/kbench.py --what mixer --dtypes bfloat16 > gpurun_out/r02n/kbench_mixer.txt 2>&1
grep -c '"finite": true' gpurun_out/r02n/kbench_mixer.txt; grep -c '"finite": false' gpurun_out/r02n/kbench_mixer.txt
timeout 300 python -m pytest tests -m gpu -x -q -k "hot_path or config2 or f32_full or mixer" 2>&1 | tail -3
timeout 200 python bench.py --model bootstapir --queries 1024 --no-accuracy 2>&1 | tail -1 > gpurun_out/r02n/bench_boots_q1024.json
cut -c1-330 gpurun_out/r02n/bench_boots_q1024.json
timeout 200 python bench.py --no-accuracy 2>&1 | tail -1 > gpurun_out/r02n/bench_warm.json
cut -c1-250 gpurun_out/r02n/bench_warm.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02n/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-accuracy > $R/gpurun_out/r02n/bench_under_rocprof.json 2>$R/gpurun_out/r02n/rocprof.err
cd $R; ls gpurun_out/r02n/prof | head; for f in $(find gpurun_out/r02n/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > gpurun_out/r02n/kernel_stats.csv; done; head -30 gpurun_out/r02n/kernel_stats.csv | cut -c1-200
find gpurun_out/r02n/prof -name '*.db' -size +20M -delete
