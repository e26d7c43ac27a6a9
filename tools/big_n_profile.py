"""profiles/r06_big_n.txt: 2^21 ... 2^24 x 2 complex64 -- us per step (wall clock, un-profiled), rocprofv3 kernel averages, PMC traffic per launch (FETCH_SIZE x 2 +
WRITE_SIZE, the guide's gfx950 correction; separate passes) and the step's fraction of the HBM roofline (32 B per sample-step / 8 TB/s).
    python3 tools/big_n_profile.py > gpurun_out/r06_big_n.txt"""
import csv, glob, os, re, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
STEPS = 100


def short(name):
    m = re.match(r"(?:void )?(?:ssfm::)?(k_[a-z_]+)<([^>]*)>", name)
    if not m:
        return name[:60]
    return f"{m.group(1)}<{m.group(2)[:44]}>"


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
    # the kernels of a step (the plan's lanes run one launch of each per step, each over one row = one polarisation): two-kernel engine k_time<MID> + k_freq<PHASE>;
    # split plans k_time<MID>, k_freq<FWD_ONLY>, k_split_mid, k_freq<INV_ONLY>.  (The other instantiations in the trace are BEGIN / END and the lane rating.)
    def in_step(name):
        if "k_split_mid" in name:
            return True
        m = re.search(r"k_time<[a-z]+, \d+, \d+, \d+, (\d+)", name)
        if m:
            return m.group(1) == "1"
        m = re.search(r"k_freq(?:_half)?<[a-z]+, \d+, \d+, \d+, (\d+)", name)
        if m:
            return m.group(1) in (("2", "5") if k > 22 else ("3",))
        return False
    print(f"   {'kernel (one launch = one lane = one polarisation)':62s} {'calls':>6s} {'avg us':>9s} {'FETCHx2 MB':>11s} {'WRITE MB':>9s} {'algorithmic MB':>15s}")
    tot_bytes, tot_us = 0.0, 0.0
    alg_launch = 2 * 8 * n                                          # one row read once + written once
    for name, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if not in_step(name):
            continue
        fe, wr = pmc["FETCH_SIZE"].get(name), pmc["WRITE_SIZE"].get(name)
        if fe is not None and wr is not None:
            tot_bytes += 2 * (2 * fe + wr)                          # two lanes
        tot_us += us
        print(f"   {short(name):62s} {calls:6d} {us:9.2f} {'' if fe is None else f'{2 * fe / 1e6:11.1f}'} {'' if wr is None else f'{wr / 1e6:9.1f}'} {alg_launch / 1e6:15.1f}")
    m = re.search(r"([\d.]+) us per step", plain)
    if m and tot_bytes:
        us = float(m.group(1))
        print(f"   a lane's kernels of one step: {tot_us:.1f} us under the profiler; PMC traffic per step (both lanes) {tot_bytes / 1e6:.0f} MB = {tot_bytes / (32.0 * n):.2f} x the algorithmic "
              f"{32 * n / 1e6:.0f} MB; {tot_bytes / us / 1e6:.2f} TB/s at the un-profiled step time; step_frac {32 * n / us * 1e6 / 8e12:.3f}", flush=True)
