# the model / config suites under the dispatch switches that select the non-default kernels (regression of the fallback paths)
cd $GRAFT_REPO_ROOT
O=gpurun_out/alt; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log)"; }
run wino16_off C2S_WINO16=0
run s2wino_off C2S_S2WINO=0
run winograd_off C2S_WINOGRAD=0
run onepass_off C2S_NORM_ONEPASS=0
run one_stream C2S_WGRAD_STREAM=0
