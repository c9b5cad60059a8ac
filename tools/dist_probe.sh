#!/bin/bash
# Diagnostic (GPU box): cost of an initialised RCCL process group on the two-stream step, world size 1 (no collective runs).
# Result (round 1): with the runtime's default 4 hardware queues the side stream collides with the main stream once RCCL
# has created its own streams (13.7 ms instead of 13.0 ms); GPU_MAX_HW_QUEUES=8 restores the overlap (bench.py sets it).
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 C2S_BENCH_FORCE_DIST=1
run() { MASTER_PORT=$1 "${@:2}" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c150-235; }
echo "process group, 8 hw queues (default of bench.py)"; run 29552 env
echo "process group, 4 hw queues"; GPU_MAX_HW_QUEUES=4 run 29553 env
echo "process group, side stream off"; C2S_WGRAD_STREAM=0 run 29554 env
echo "no process group"; C2S_BENCH_FORCE_DIST=0 run 29555 env
