set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_exp; mkdir -p $O
b() { name=$1; envs=$2; shift; shift; env $envs timeout -k 10 150 python bench.py --no-cpu-baseline "$@" > $O/c_$name.json 2> $O/c_$name.err || true; echo "$name $(python -c "import json;print(json.load(open('$O/c_$name.json'))['ms_per_step'])" 2>/dev/null || tail -2 $O/c_$name.err)"; }
for i in 1 2; do
b utae$i X=1 --steps 40 --warmup 5
b tu$i X=1 --model timeunet --batch 8 --T 61 --steps 15 --warmup 3
b tu_noflush$i C2S_WGRAD_FLUSH_POSITIONS=16777216 --model timeunet --batch 8 --T 61 --steps 15 --warmup 3
b tu_flush19_$i C2S_WGRAD_FLUSH_POSITIONS=524288 --model timeunet --batch 8 --T 61 --steps 15 --warmup 3
b wtae$i X=1 --model wtae --steps 40 --warmup 5
b wtae_flush19_$i C2S_WGRAD_FLUSH_POSITIONS=524288 --model wtae --steps 40 --warmup 5
done
b c5 X=1 --batch 8 --T 48 --size 256 --steps 5 --warmup 2
