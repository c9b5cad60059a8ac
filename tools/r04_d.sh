set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== compute-only, 16-px tiles"; C2S_DIAG_LIB=tools/_diag/libs/libc2s_computeonly.so C2S_LTAE_REG8=0 python tools/ltae_diag.py 2>&1 | tail -1
echo "== compute-only, 8-px tiles"; C2S_DIAG_LIB=tools/_diag/libs/libc2s_computeonly.so C2S_LTAE_STAGGER=0 python tools/ltae_diag.py 2>&1 | tail -1
