import sys, time
sys.path.insert(0, "/root/repo")
import bench, subprocess, os, shutil, tempfile
t=time.time()
prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
out = tempfile.mkdtemp(prefix="mwpmc_", dir="/tmp")
env = dict(os.environ, MW_BENCH_CHILD="1", TMPDIR="/tmp")
r = subprocess.run([prof, "--pmc", "FETCH_SIZE", "WRITE_SIZE", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                sys.executable, "/root/repo/bench.py", "--config", "hallway", "--envs-per-gpu", "4096", "--steps", "6", "--warmup", "2",
                "--no-cpu-baseline", "--no-parity-check", "--no-also"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=280)
print("rc", r.returncode, time.time()-t)
print(r.stdout[-600:]); print(r.stderr[-1500:])
for root, _, files in os.walk(out):
    for f in files: print(os.path.join(root,f), os.path.getsize(os.path.join(root,f)))
