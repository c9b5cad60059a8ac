set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_prof; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_2s.json 2> $O/prof_2s.err
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_1s.json 2> $O/prof_1s.err
echo traces done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/st_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/st_fetch.json 2> $O/st_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/st_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/st_write.json 2> $O/st_write.err
python tools/step_traffic.py $O/st_fetch $O/st_write $O/step_traffic.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/pmc_traffic.py run > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/pmc_traffic.py run > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py parse $O/pmc_fetch $O/pmc_write > $O/pmc_parse.log 2>&1 || tail -5 $O/pmc_parse.log
cp profiles/dominant_kernel_traffic.json $O/
find $O -name '*kernel_trace.csv' -delete
find $O -name '*counter_collection.csv' -size +30M -delete
du -sh $O
