# round 2, first GPU session: new multi-GPU tests, default bench line, SQ counters behind the VALU claim, microbench tables
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O; cd $R
python -c "import tsta_amd; print(tsta_amd.build_info(), tsta_amd.source_hash())" > $O/build_info.txt 2>&1
echo "(multi tests ran in the previous call)"

timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 1500 $O/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/prof_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_sq.json 2> $O/pmc_sq.err
echo "sq done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/prof_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-related --no-verify > $O/pmc_sq2.json 2> $O/pmc_sq2.err || echo "sq2 failed (counter set)"
echo "sq2 done"
for t in cell_microbench valu_microbench step_microbench; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$t tools/$t.hip && timeout -k 10 200 /tmp/$t > $O/$t.txt 2>&1 || echo "$t failed"
done
echo "microbench done"
