"""One-off: bench line in short form (v2 + the container_v3 reading)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu"] + sys.argv[1:], capture_output=True, text=True)
for l in out.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print("v2: enc %.3f ms %.1f GB/s  dec %.1f GB/s  pipelined %.1f  | v3: %s" % (d["ms_per_step"], d["value"] / 1e3, d["decode_MBps"] / 1e3,
              d.get("pipelined", {}).get("value", 0) / 1e3, d.get("container_v3")))
if out.returncode:
    print(out.stderr[-1500:])
