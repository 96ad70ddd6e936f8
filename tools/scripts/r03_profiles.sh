# round 3 profiles: default bench line (with configs #2/#3/#4/5t), rocprofv3 kernel stats of the headline command, PMC traffic
# (separate passes), SQ counters, the configs with their kernel stats.  Run on the GPU box: bash tools/scripts/r03_profiles.sh <tag>  (profiles/r03k_* = this script on the final build of round 3, condensed by tools/collect_profiles.py)
set -e
export TMPDIR=/tmp
TAG=${1:-r03k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -c "import tsta_amd; print(tsta_amd.build_info(), tsta_amd.source_hash())" > $O/build_info.txt 2>&1
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 300 $O/bench_default.json
LEAN="--no-cpu-baseline --no-generic --no-related --no-verify --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 2 --warmup 1 $LEAN > $O/bench_prof.json 2> $O/prof.err
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 1 --warmup 0 $LEAN > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 1 --warmup 0 $LEAN > $O/pmc_write.json 2> $O/pmc_write.err
echo "traffic done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/prof_sq -- python3 bench.py --steps 1 --warmup 0 $LEAN > $O/pmc_sq.json 2> $O/pmc_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/prof_sq2 -- python3 bench.py --steps 1 --warmup 0 $LEAN > $O/pmc_sq2.json 2> $O/pmc_sq2.err
echo "sq done"
TSTA_BENCH_NO_CPU=1 timeout -k 10 400 python bench_configs.py 2 3 4 5t 5m 5s 5h > $O/configs.jsonl 2> $O/configs.err
echo "configs done"; cut -c1-160 $O/configs.jsonl
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 bench_configs.py 2 3 4 5t 5m > $O/configs_prof.jsonl 2> $O/prof_cfg.err
echo "config stats done"
