set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_x
for v in none side; do C2S_EXP_SKIP=$v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r3_x/s_$v.json 2>/dev/null; echo "$v $(python -c "import json;print(json.load(open('gpurun_out/r3_x/s_$v.json'))['ms_per_step'])")"; done
C2S_EXP_SKIP=side C2S_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_x/prof_main -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_x/prof_main.json 2> gpurun_out/r3_x/prof_main.err
find gpurun_out/r3_x -name '*kernel_trace.csv' -delete
