mkdir -p gpurun_out/wide
{
for WT in 1 0; do
  echo "== TSTA_WALK_TABLE=$WT  4 x 30 kbp, 10 %"
  TSTA_WALK_TABLE=$WT timeout -k 10 200 python tests/tools/msa_wide_timing.py 4 30000 0.1 2>&1 | head -1
  echo "== TSTA_WALK_TABLE=$WT  3 x 100 kbp, 10 %"
  TSTA_WALK_TABLE=$WT timeout -k 10 300 python tests/tools/msa_wide_timing.py 3 100000 0.1 2>&1 | head -1
done
} > gpurun_out/wide/wide_walk.txt 2>&1
cat gpurun_out/wide/wide_walk.txt
