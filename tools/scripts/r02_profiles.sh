# round 2 profiles: default bench line, rocprofv3 kernel stats of the same command, PMC traffic (separate passes), SQ counters,
# the other configs with their kernel stats
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 400 $O/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generic --no-related --no-verify > $O/bench_prof.json 2> $O/prof.err
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_write.json 2> $O/pmc_write.err
echo "traffic done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/prof_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_sq.json 2> $O/pmc_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/prof_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_sq2.json 2> $O/pmc_sq2.err
echo "sq done"
timeout -k 10 400 python bench_configs.py 2 3 4 5t 5m 5s 5h > $O/configs.jsonl 2> $O/configs.err
echo "configs done"; cut -c1-200 $O/configs.jsonl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 bench_configs.py 2 3 4 5t 5m > $O/configs_prof.jsonl 2> $O/prof_cfg.err
echo "config stats done"
