# round 3, second call: HBM traffic of the configs' dominant kernels (separate FETCH_SIZE / WRITE_SIZE passes over bench_configs.py 2 3 4 5t);
# tools/collect_profiles.py condenses them into profiles/traffic_configs.json.  Run on the GPU box: bash tools/scripts/r03_profiles_cfg_traffic.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-r03k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_cfg_fetch -- python3 bench_configs.py 2 3 4 5t > $O/configs_fetch.jsonl 2> $O/prof_cfg_fetch.err
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_cfg_write -- python3 bench_configs.py 2 3 4 5t > $O/configs_write.jsonl 2> $O/prof_cfg_write.err
echo "config traffic done"
