"""The drop-in CLI over the bench's own captures (bench.ook_batches: every third a protocol-valid transmission), as bench.py's
`dropin` leg runs it but without a parent process on the GPU, and with the flow's trace:
    python tools/cli_bench_files.py [N] [NAME=VALUE ... environment of the CLI, e.g. RTL433_HIP_PREFILTER=1]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
extra = dict(a.split('=', 1) for a in sys.argv[2:])
d = "/dev/shm/r433_cli_files"
os.makedirs(d, exist_ok=True)
host = bench.ook_batches(0, n, 8)
names = []
for k in range(n):
    names.append(f"s{k:05d}_433.92M_250k.cu8")
    host[k].tofile(os.path.join(d, names[-1]))
args = [a for f in names for a in ("-r", f)] + ["-F", "json", "-M", "level", "-K", "FILE"]
cli = os.path.join(bench.ROOT, "dropin", "_build", "rtl_433_hip")
for rep in range(4):
    t0 = time.perf_counter()
    p = subprocess.run([cli] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RTL433_HIP_TRACE="1" if rep == 3 else "0", **({"RTL433_HIP_DEBUG": "32"} if rep == 3 else {}), **extra))
    print(f"{extra} rep {rep}: {(time.perf_counter() - t0) * 1e3:.0f} ms, {p.stdout.count(10)} lines")
print(p.stderr.decode(errors="replace"))
