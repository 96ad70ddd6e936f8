set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01e; mkdir -p $O; cd $R
python bench_configs.py 2 3 4 5t 5m 5s 5h > $O/configs.jsonl 2> $O/configs.err
echo "configs done"; cut -c1-260 $O/configs.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 bench_configs.py 2 3 4 5t 5m > $O/configs_prof.jsonl 2> $O/prof_cfg.err
echo "stats done"
