"""One-off: kernel times of a version-3 encode / decode of the bench batch (FQZ_DBG_SERIAL=1 for standalone times)."""
import os, sys, json, subprocess
env = dict(os.environ)
out = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--container", "3", "--profile", "1", "--inflight", "0", "--steps", "5"], env=env, capture_output=True, text=True)
for l in out.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print("encode %.3f ms  decode %.1f GB/s  ratio %.3f  roundtrip %s" % (d["ms_per_step"], d["decode_MBps"] / 1e3, d["ratio"], d["roundtrip_bit_exact"]))
        print(" enc:", {k: v for k, v in d["kernel_ms"].items() if v > 0.08})
        print(" dec:", {k: v for k, v in d["decode_kernel_ms"].items() if v > 0.08})
if out.returncode:
    print(out.stderr[-2000:])
