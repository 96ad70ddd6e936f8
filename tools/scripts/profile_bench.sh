set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01e; mkdir -p $O; cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 600 $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generic > $O/bench_prof.json 2> $O/prof.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic > $O/pmc_write.json 2> $O/pmc_write.err
echo "write done"
