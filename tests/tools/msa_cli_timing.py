"""One-off (GPU box): wall time of the TSTA_msa drop-in on the README example next to the reference binary, with the
drop-in's own breakdown (TSTA_MSA_TIMING) and a few runtime environment knobs."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
IN = os.path.join(ROOT, "tests", "golden", "msa_seq1.fa")
def run(cmd, env=None, reps=5):
    e = dict(os.environ); e["TSTA_MSA_TIMING"] = "1"; e.update(env or {})
    ts, last = [], ""
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, env=e)
        ts.append(time.perf_counter() - t0)
        last = r.stderr.decode().strip().split("\n")[-1]
    print("%-44s wall min %.3f median %.3f s | %s" % (" ".join("%s=%s" % kv for kv in (env or {}).items()) or os.path.basename(cmd[0]), min(ts), sorted(ts)[len(ts) // 2], last), flush=True)
ours = [os.path.join(ROOT, "tsta_amd", "bin", "TSTA_msa"), "-T", "4", "-i", IN, "-o", "/tmp/o.txt"]
ref = [os.path.join(ROOT, "oracle", "_ref", "TSTA_msa"), "-T", "64", "-i", IN, "-o", "/tmp/o2.txt"]
subprocess.run(["gcc", "-O2", "-o", "/tmp/hsa_init_time", os.path.join(ROOT, "tests/tools/hsa_init_time.c"), "-I/opt/rocm/include", "-L/opt/rocm/lib",
                "-lhsa-runtime64", "-lamdhip64", "-D__HIP_PLATFORM_AMD__", "-Wl,-rpath,/opt/rocm/lib"], check=True)
for _ in range(3):
    print(subprocess.run(["/tmp/hsa_init_time"], capture_output=True).stdout.decode().strip(), flush=True)
run(ours)
for env in ({"HSA_ENABLE_SDMA": "0"}, {"HSA_ENABLE_INTERRUPT": "0"}, {"HSA_ENABLE_SDMA": "0", "HSA_ENABLE_INTERRUPT": "0"}, {"HIP_HOST_COHERENT": "0"},
            {"AMD_DIRECT_DISPATCH": "0"}, {"HSA_NO_SCRATCH_RECLAIM": "1"}, {"HSA_TOOLS_LIB": ""}, {"ROCR_VISIBLE_DEVICES": "0", "HSA_ENABLE_SDMA": "0"}):
    run(ours, env)
run(ref)
