# final check of the round: full GPU suite, configs with their kernel stats
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 400 python bench_configs.py 2 3 4 5t 5m > $O/configs.jsonl 2> $O/configs.err
echo "configs done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 bench_configs.py 2 3 4 5t 5m > $O/configs_prof.jsonl 2> $O/prof_cfg.err
echo "config stats done"
