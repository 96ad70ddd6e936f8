# round 4 profiles, ONE call on the GPU box: bash tools/scripts/r04_profiles.sh <tag>
#   default bench line (e2e, mixed alphabet, configs #2/#3/#4/5t), rocprofv3 kernel stats of the headline command, PMC traffic (separate
#   FETCH_SIZE / WRITE_SIZE passes), SQ counters, the configs with their kernel stats and PMC traffic, side timings (K4e, wave log,
#   e2e, microbench).  tools/collect_profiles.py <dir> <tag> condenses it into profiles/<tag>_*; build_info.txt ties it to the build.
set -e
export TMPDIR=/tmp
TAG=${1:-r04z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -c "import tsta_amd; print(tsta_amd.build_info())" > $O/build_info.txt 2>&1
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; python tools/scripts/show_bench.py $O/bench_default.json | head -12
LEAN="--no-cpu-baseline --no-generic --no-related --no-verify --no-configs --no-e2e"
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
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_cfg_fetch -- python3 bench_configs.py 2 3 4 5t > $O/configs_fetch.jsonl 2> $O/prof_cfg_fetch.err
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_cfg_write -- python3 bench_configs.py 2 3 4 5t > $O/configs_write.jsonl 2> $O/prof_cfg_write.err
echo "config traffic done"
{
  echo "==== build"; cat $O/build_info.txt
  echo "==== tests/tools/k4e_timing.py"; timeout -k 10 200 python tests/tools/k4e_timing.py 2>&1 | tail -3
  echo "==== tests/tools/msa_n_timing.py"; timeout -k 10 200 python tests/tools/msa_n_timing.py 2>&1 | tail -8
  echo "==== tests/tools/msa_cli_timing.py"; timeout -k 10 200 python tests/tools/msa_cli_timing.py 2>&1 | tail -8
  echo "==== tests/tools/e2e_timing.py"; timeout -k 10 200 python tests/tools/e2e_timing.py 2>&1 | grep -v amdgpu.ids
  echo "==== tests/tools/wave_log.py (P=8192)"; timeout -k 10 200 python tests/tools/wave_log.py 2>&1 | grep -v "amdgpu.ids\|\[tsta\]"
  echo "==== tests/tools/wave_log.py (P=4096)"; P=4096 timeout -k 10 200 python tests/tools/wave_log.py 2>&1 | grep -v "amdgpu.ids\|\[tsta\]"
  echo "==== tests/tools/mixed_alphabet_timing.py"; timeout -k 10 200 python tests/tools/mixed_alphabet_timing.py 2>&1 | grep -v amdgpu.ids
  echo "==== tests/tools/walk_table_timing.py"; timeout -k 10 200 python tests/tools/walk_table_timing.py 2>&1 | grep -v amdgpu.ids
} > $O/side_timings.txt 2>&1
./tools/cell_microbench > $O/cell_microbench.txt 2>&1 || true
echo "side timings done"
