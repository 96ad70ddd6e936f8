# one-off: where the wall time of the TSTA_msa drop-in goes on the README example (run on the GPU box)
cd $GRAFT_REPO_ROOT
IN=tests/golden/msa_seq1.fa
t() { local s=$(date +%s.%N); "$@" > /dev/null 2>/tmp/err.txt; local e=$(date +%s.%N); echo "$(echo "$e - $s" | bc) s : $* ; $(tail -1 /tmp/err.txt)"; }
gcc -O2 -o /tmp/init_time tests/tools/init_time.c -Ltsta_amd/lib -ltsta_hip -Wl,-rpath,$PWD/tsta_amd/lib
for i in 1 2 3; do /tmp/init_time; done
export TSTA_MSA_TIMING=1
for i in 1 2 3; do t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- HIP_VISIBLE_DEVICES=0"; for i in 1 2; do HIP_VISIBLE_DEVICES=0 t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- ROCR_VISIBLE_DEVICES=0"; for i in 1 2; do ROCR_VISIBLE_DEVICES=0 t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- HSA_ENABLE_SDMA=0"; for i in 1 2; do HSA_ENABLE_SDMA=0 t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- GPU_MAX_HW_QUEUES=1"; for i in 1 2; do GPU_MAX_HW_QUEUES=1 t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- HIP_FORCE_DEV_KERNARG=1"; for i in 1 2; do HIP_FORCE_DEV_KERNARG=1 t tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt; done
echo "--- reference"; for i in 1 2 3; do t oracle/_ref/TSTA_msa -T 64 -i $IN -o /tmp/o2.txt; done
echo "--- strace -c (syscall time)"; strace -f -c -o /tmp/strace.txt tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt > /dev/null 2>&1; head -15 /tmp/strace.txt
echo "--- ltrace-ish: perf of dlopen"; LD_DEBUG=statistics tsta_amd/bin/TSTA_msa -T 4 -i $IN -o /tmp/o.txt 2>&1 | grep -i "total startup\|relocation" | head -5
nproc; rocminfo 2>/dev/null | grep -c "Device Type:.*GPU"
