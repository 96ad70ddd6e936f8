import os, sys, subprocess, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
from tests.golden.make_poa_golden import family, write_multi_fasta
for (n,l,r) in [(40,5000,0.05),(12,30000,0.03),(5,100000,0.03)]:
    fa="/tmp/m.fa"; write_multi_fasta(fa, family(4242,n,l,r))
    t=time.perf_counter()
    p=subprocess.run([os.path.join(ROOT,"tsta_amd/bin/TSTA_msa"),"-i",fa,"-o","/tmp/o.txt"],capture_output=True,env=dict(os.environ,TSTA_MSA_TIMING="1"))
    print(n,l,"%.2f s"%(time.perf_counter()-t), p.stderr.decode().strip().split("\n")[-1])
