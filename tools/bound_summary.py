"""Per-kernel bound summary from the committed rocprofv3 evidence: for every kernel that takes more than 2 % of a
step, the numbers that say what limits it.

usage: python tools/bound_summary.py r03      (after tools/prof.sh, tools/pmc_sq.sh and tools/collect_profiles.py r03)
reads   gpurun_out/pmc_sq/summary.txt                                   SQ counters per launch (two --pmc passes)
        profiles/<round>/rocprofv3_kernel_stats_bench_steps5.csv         average duration per launch
        profiles/<round>/pmc_hbm_traffic.json                            HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes)
writes  profiles/<round>/bound_summary.txt, profiles/<round>/pmc_sq_summary.txt

Definitions (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 = 1024 SIMDs; a wave64 VALU instruction takes 2 cycles of its SIMD; SQ_WAVE_CYCLES,
SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles; clock taken as 2.4 GHz, the chip's maximum - under load it runs lower, so the
utilisations below are lower bounds):
  cycles            = average duration x 2.4 GHz
  VALU utilisation  = SQ_INSTS_VALU x 2 / (1024 x cycles)
  waves per SIMD    = SQ_WAVE_CYCLES x 4 / (1024 x cycles)            (resident, averaged over the kernel)
  parked share      = SQ_WAIT_ANY / SQ_WAVE_CYCLES                    (waves waiting at s_waitcnt / barriers)
  issue-stall share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  bytes per VMEM    = HBM bytes / (SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR)   (a full wave access moves 1024 bytes)
  LDS conflicts     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  HBM rate          = HBM bytes / duration                            (peak 8 TB/s spec, ~6.3 TB/s achievable)
"""
import csv, json, os, re, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", rnd)
sq_src = os.path.join(root, "gpurun_out", "pmc_sq", "summary.txt")
if os.path.exists(sq_src):
    shutil.copy(sq_src, os.path.join(dst, "pmc_sq_summary.txt"))
sq = {}
for ln in open(os.path.join(dst, "pmc_sq_summary.txt")):
    name = ln[:40].split("(")[0].strip()
    sq[name] = {k: float(v) for k, v in re.findall(r"(SQ_[A-Z_]+)=([0-9.e+]+)", ln)}
dur = {}
for r in csv.DictReader(open(os.path.join(dst, "rocprofv3_kernel_stats_bench_steps5.csv"))):
    dur[r["Name"].split("(")[0]] = (float(r["AverageNs"]), float(r["Percentage"]))
hbm = {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(os.path.join(dst, "pmc_hbm_traffic.json")))["kernels"].items()}
CLK = 2.4e9
rows = []
for k, (ns, pct) in sorted(dur.items(), key=lambda x: -x[1][0] * 0 - x[1][1]):
    c = sq.get(k)
    if not c or pct < 2.0 or k.startswith("void at::") or k.startswith("__amd"):
        continue
    cyc = ns * 1e-9 * CLK
    valu = c.get("SQ_INSTS_VALU", 0) * 2 / (1024 * cyc)
    waves = c.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cyc)
    parked = c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1))
    stall = c.get("SQ_WAIT_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1))
    vmem = c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    bpv = hbm.get(k, 0) / vmem if vmem else 0
    ldsc = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0
    ldsbusy = c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc)  # LDS-array cycles per CU cycle
    rate = hbm.get(k, 0) / (ns * 1e-9) / 1e12
    # the limiter, read off the numbers above
    if rate > 3.5:
        lim = "HBM bandwidth (%.1f TB/s of ~6.3 achievable)" % rate
    elif ldsbusy > 0.45:
        lim = "LDS pipeline (%.0f %% of the LDS cycles busy, %.0f %% of them bank conflicts)" % (100 * ldsbusy, 100 * ldsc)
    elif valu > 0.6:
        lim = "VALU issue (%.0f %%)" % (100 * valu)
    elif parked > 0.5 and bpv and bpv < 400:
        lim = "memory latency behind narrow accesses (%.0f %% of wave time parked, %.0f B per vector-memory instruction of 1024)" % (100 * parked, bpv)
    elif parked > 0.5:
        lim = "latency: %.0f %% of wave time parked at waits / barriers, VALU %.0f %%, %.1f waves per SIMD" % (100 * parked, 100 * valu, waves)
    else:
        lim = "mixed: VALU %.0f %%, parked %.0f %%, issue stalls %.0f %%" % (100 * valu, 100 * parked, 100 * stall)
    rows.append((k, ns / 1e6, pct, valu, waves, parked, stall, bpv, ldsc, ldsbusy, rate, lim))
out = [__doc__.split("Definitions")[0].strip().splitlines()[0], "",
       "%-16s %7s %5s %6s %6s %7s %6s %7s %6s %6s %6s" % ("kernel", "ms", "%", "VALU", "waves", "parked", "stall", "B/VMEM", "LDScf", "LDSbz", "TB/s")]
for r in rows:
    out.append("%-16s %7.3f %5.1f %5.0f%% %6.1f %6.0f%% %5.0f%% %7.0f %5.0f%% %5.0f%% %6.2f" % (r[0][:16], r[1], r[2], 100 * r[3], r[4], 100 * r[5], 100 * r[6], r[7], 100 * r[8], 100 * r[9], r[10]))
out.append("")
for r in rows:
    out.append("%-16s limited by %s" % (r[0][:16], r[11]))
out.append("")
out.append("Definitions" + __doc__.split("Definitions")[1].rstrip())
open(os.path.join(dst, "bound_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
