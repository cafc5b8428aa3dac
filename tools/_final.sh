R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/final_run.sh r06 full 2>&1 | tail -40
bash tools/run_profiles.sh r06_x3 bf16x3 all 2>&1 | tail -5
ls -la gpurun_out | tail -40
