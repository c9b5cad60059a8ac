set -e
O=gpurun_out/gx; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_ltae_paths_gpu.py -x -q -m gpu -k "ltae" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -3 $O/t.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -- python tools/ltae_bench.py --no-attn --reps 3 > $O/b1.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s2 -- python tools/ltae_bench.py --reps 3 > $O/b2.txt 2>&1
for f in $O/s1/*/*kernel_stats.csv $O/s2/*/*kernel_stats.csv; do echo $f; grep -i "ltae" $f | cut -d, -f1-4 | cut -c1-150; done
