"""lqcov_run_files in THIS process (python: torch's HIP runtime is loaded first when torch is imported) on the files of tools/gpu_exe.sh"""
import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
if os.environ.get("WITH_TORCH", "1") == "1":
    import torch  # noqa: F401
from longqc_amd import api
argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", "-t", "8", "/dev/shm/all.fa", "/dev/shm/sub.fq"]
p, _, _ = api.parse_args(argv)
eng = api.Engine(p, device=0)
t0 = time.time()
eng.run_files("/dev/shm/all.fa", "/dev/shm/sub.fq", out="/dev/shm/out_py.tsv", err="/dev/shm/err_py.log")
dt = time.time() - t0
eng.close()
print("in-process run_files (torch %s): %.2f s" % (os.environ.get("WITH_TORCH", "1"), dt)); os.system("python tools/check_rows.py cfg3 /dev/shm/out_py.tsv")
print("\n".join(l for l in open("/dev/shm/err_py.log").read().splitlines() if "mapped" in l))
