set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r02q
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tests.golden.make_poa_golden import family, write_multi_fasta
write_multi_fasta("/tmp/msa_wide.fa", family(777, 3, 100000, 0.1))
PY
for df in 1 0; do
  TSTA_POA_DATAFLOW=$df rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02q/prof_wide_$df -- tsta_amd/bin/TSTA_msa -i /tmp/msa_wide.fa -o /tmp/w_$df.txt > /dev/null 2>&1
  f=$(ls -t gpurun_out/r02q/prof_wide_$df/*/*_kernel_stats.csv | head -1)
  echo "dataflow=$df"; head -6 $f | cut -d, -f1-4 | cut -c1-120
done
md5sum /tmp/w_1.txt /tmp/w_0.txt
