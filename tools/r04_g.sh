set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
echo "default:"; python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
echo "direct s2 wgrad:"; C2S_S2WINO_WGRAD=0 python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
