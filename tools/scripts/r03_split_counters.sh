# round 3: SQ counters of config #3's fill (psa_trace_split_kernel) and config #4's (poa_dataflow_kernel): how busy the chip is
# when the bound is a chain of dependent steps.  Run on the GPU box: bash tools/scripts/r03_split_counters.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-r03k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/prof_cfg_sq -- python3 bench_configs.py 3 4 > $O/configs_sq.jsonl 2> $O/prof_cfg_sq.err
TSTA_BENCH_NO_CPU=1 TSTA_BENCH_NO_VERIFY=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $O/prof_cfg_sq2 -- python3 bench_configs.py 3 4 > $O/configs_sq2.jsonl 2> $O/prof_cfg_sq2.err
echo "config counters done"
