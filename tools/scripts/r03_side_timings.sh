# round 3: the timings DESIGN.md quotes outside the bench line, on the final build, into one text file (profiles/r03k_side_timings.txt):
# TSTA_msa start-up breakdown next to the reference, wide MSAs (4 x 30 kbp, 3 x 100 kbp: K4d against K4), the K4e path (per fill and
# end to end on the README example with three N's).  Run on the GPU box: bash tools/scripts/r03_side_timings.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r03k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
F=$O/side_timings.txt
python -c "import tsta_amd; print(tsta_amd.build_info())" > $F 2>&1
echo "==== tests/tools/msa_cli_timing.py" >> $F; timeout -k 10 200 python tests/tools/msa_cli_timing.py >> $F 2>&1
echo "==== tests/tools/msa_wide_timing.py 4 30000 0.1" >> $F; timeout -k 10 300 python tests/tools/msa_wide_timing.py 4 30000 0.1 >> $F 2>&1
echo "==== tests/tools/msa_wide_timing.py 3 100000 0.1" >> $F; timeout -k 10 400 python tests/tools/msa_wide_timing.py 3 100000 0.1 >> $F 2>&1
echo "==== tests/tools/k4e_timing.py" >> $F; timeout -k 10 200 python tests/tools/k4e_timing.py >> $F 2>&1
echo "==== tests/tools/msa_n_timing.py" >> $F; timeout -k 10 200 python tests/tools/msa_n_timing.py >> $F 2>&1
echo "side timings done"; tail -30 $F | cut -c1-220
