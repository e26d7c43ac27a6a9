"""profiles/r06_big_n.txt: 2^21 ... 2^24 x 2 complex64 -- us per step (wall clock, un-profiled), rocprofv3 kernel averages, PMC traffic per launch (FETCH_SIZE x 2 +
WRITE_SIZE, the guide's gfx950 correction; separate passes) and the step's fraction of the HBM roofline (32 B per sample-step / 8 TB/s).
    python3 tools/big_n_profile.py > gpurun_out/r06_big_n.txt"""
import csv, glob, os, re, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
STEPS = 20


def short(name):
    m = re.match(r"(?:void )?(?:ssfm::)?(k_[a-z_]+)<([^>]*)>", name)
    if not m:
        return name[:40]
    return f"{m.group(1)}<{m.group(2)[:28]}>"


def run(cmd, **kw):
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, **kw)


for k in (21, 22, 23, 24):
    plain = run([sys.executable, "tools/big_n_run.py", str(k), str(STEPS)]).stdout.strip()
    print("== " + plain, flush=True)
    d = f"/tmp/bign_{k}"
    subprocess.run(["rm", "-rf", d])
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d + "_kt", "--", "python3", "tools/big_n_run.py", str(k), str(STEPS)])
    stats = {}
    for f in glob.glob(d + "_kt/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            stats[row["Name"]] = (int(row["Calls"]), float(row["TotalDurationNs"]) / int(row["Calls"]) / 1e3)
    pmc = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        run(["rocprofv3", "--pmc", C, "--output-format", "csv", "-d", f"{d}_{C}", "--", "python3", "tools/big_n_run.py", str(k), "4"])
        acc = defaultdict(list)
        for f in glob.glob(f"{d}_{C}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == C:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]) * 1024)
        pmc[C] = {kk: sum(v) / len(v) for kk, v in acc.items()}
    n = 1 << k
    print(f"   {'kernel':52s} {'calls':>6s} {'avg us':>9s} {'FETCHx2 MB':>11s} {'WRITE MB':>9s} {'algorithmic MB':>15s}")
    tot_bytes = 0.0
    for name, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if calls < STEPS or not ("k_time" in name or "k_freq" in name or "k_split" in name):
            continue
        fe, wr = pmc["FETCH_SIZE"].get(name), pmc["WRITE_SIZE"].get(name)
        per_step = calls / (2.0 * STEPS)                      # launches per step (two runs of STEPS steps)
        alg = 2 * 2 * 8 * n / max(per_step, 1e-9) if "k_split" not in name else 2 * 2 * 8 * n
        if fe is not None and wr is not None:
            tot_bytes += per_step * (2 * fe + wr)
        print(f"   {short(name):52s} {calls:6d} {us:9.2f} {'' if fe is None else f'{2 * fe / 1e6:11.1f}'} {'' if wr is None else f'{wr / 1e6:9.1f}'} {alg / 1e6:15.1f}")
    m = re.search(r"([\d.]+) us per step", plain)
    if m and tot_bytes:
        us = float(m.group(1))
        print(f"   traffic per step {tot_bytes / 1e6:.0f} MB = {tot_bytes / (32.0 * n):.2f} x the algorithmic {32 * n / 1e6:.0f} MB; {tot_bytes / us / 1e6:.2f} TB/s at the un-profiled step time", flush=True)
