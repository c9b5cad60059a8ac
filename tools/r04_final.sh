# Round-4 measurement pass (GPU box): tests, smoke, bench lines for the BASELINE configs, rocprofv3 kernel stats (two streams,
# one stream, TimeUNet), the per-launch extract of the dominant kernel, PMC traffic of the step / the dominant kernel / the L-TAE
# block.  Results land in gpurun_out/r4_final; the summaries to keep are copied to profiles/ by hand.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_final; mkdir -p $O
if [ "$1" != "notests" ]; then
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
fi
python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_timeunet_b8_t61.json 2> $O/tu.err
python bench.py --model wtae --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_wtae.json 2> $O/wt.err
python bench.py --batch 8 --T 48 --size 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_utae_b8_t48_256.json 2> $O/c5.err
python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_graph.json 2> $O/gr.err
for f in bench_timeunet_b8_t61 bench_wtae bench_utae_b8_t48_256 bench_graph; do echo "$f $(python -c "import json;print(json.load(open('$O/$f.json'))['ms_per_step'])")"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_2s.json 2> $O/prof_2s.err
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_1s.json 2> $O/prof_1s.err
python tools/dominant_extract.py $O/prof_1s $O/utae_dominant_kernel.csv
python tools/trace_gaps.py $O/prof_2s 4 > $O/trace_gaps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tu -- python bench.py --model timeunet --batch 8 --T 61 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_tu.json 2> $O/prof_tu.err
find $O -name '*kernel_trace.csv' -delete
# HBM traffic: the step by kernel, the dominant kernel (calibrated), the L-TAE block at the TimeUNet shape
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/st_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/st_fetch.json 2> $O/st_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/st_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/st_write.json 2> $O/st_write.err
python tools/step_traffic.py $O/st_fetch $O/st_write $O/step_traffic.csv | tail -3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/pmc_traffic.py run > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/pmc_traffic.py run > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py parse $O/pmc_fetch $O/pmc_write > $O/pmc_parse.log 2>&1 || tail -5 $O/pmc_parse.log
cp profiles/dominant_kernel_traffic.json $O/
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/lt_fetch -- python tools/ltae_bench.py --no-attn --reps 2 > $O/lt_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/lt_write -- python tools/ltae_bench.py --no-attn --reps 2 > $O/lt_w.log 2>&1
python tools/step_traffic.py $O/lt_fetch $O/lt_write $O/ltae_traffic.csv 2 > $O/lt_parse.log 2>&1 || tail -3 $O/lt_parse.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lt_stats -- python tools/ltae_bench.py --no-attn --reps 3 > $O/ltae_bench.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/mfma -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/mfma.json 2> $O/mfma.err
python tools/mfma_util.py $O/mfma $O/mfma_util.csv > /dev/null
timeout -k 10 300 python tools/tile_bench.py > $O/tile_bench.txt 2>&1 || tail -5 $O/tile_bench.txt
find $O -name '*kernel_trace.csv' -delete
find $O -name '*counter_collection.csv' -size +30M -delete
du -sh $O
