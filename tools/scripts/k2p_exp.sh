cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c
for cfg in "3 64" "3 256" "2 256" "3 192"; do set -- $cfg
  TSTA_K2P_PER_CU=$1 TSTA_TRACE_T=$2 timeout -k 10 200 python bench_configs.py 5t 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('per_cu=$1 T=$2 kernel_ms=%.1f wall=%.1f first=%.1f' % (j['kernel_ms'], j['wall_ms'], j['wall_ms_first']))"
done
