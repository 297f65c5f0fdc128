set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04final; mkdir -p $OUT
R=$PWD
timeout 400 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json; cut -c1-200 $OUT/bench_bf16.json
bash tools/final_artefacts_r04.sh prof r04final
