set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_stamps; mkdir -p $O
timeout -k 10 600 python tools/ltae_reg_stamps.py > $O/fwd.txt 2>&1 || { tail -20 $O/fwd.txt; exit 1; }
cat $O/fwd.txt | tail -15
timeout -k 10 600 python tools/ltae_regbwd_stamps.py > $O/bwd.txt 2>&1 || { tail -20 $O/bwd.txt; exit 1; }
cat $O/bwd.txt | tail -25
