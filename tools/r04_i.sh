cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "f32-timeunet_train" 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -5
