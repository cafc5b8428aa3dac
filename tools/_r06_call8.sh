R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_model_gpu.py -q -x -k "several_steps" 2>&1 | grep -E "^E |assert|Error|passed|failed" | cut -c1-400 | head -30
