set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline $1 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for t in 0 1 2 3 4 5 6 8; do echo "TAIL_OPS=$t: $(C2S_WGRAD_TAIL_OPS=$t run) $(C2S_WGRAD_TAIL_OPS=$t run)"; done
