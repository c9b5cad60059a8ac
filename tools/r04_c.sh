set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "ltae or timeunet or C3 or both_conv" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
C2S_LTAE_REG8=0 timeout -k 10 200 python tools/ltae_bench.py --no-attn > $O/ltae_reg16.txt 2>&1
timeout -k 10 200 python tools/ltae_bench.py --no-attn > $O/ltae_reg8.txt 2>&1
timeout -k 10 200 python tools/ltae_bench.py > $O/ltae_reg8_attn.txt 2>&1
tail -2 $O/ltae_reg16.txt; tail -2 $O/ltae_reg8.txt; tail -2 $O/ltae_reg8_attn.txt
