"""Turn a rocprofv3 --kernel-trace --stats result (rocpd .db or *kernel_stats.csv) into a compact
markdown table for profiles/.  Usage: python tools/summarize_prof.py <dir-or-db> [steps] > profiles/x.md"""
import csv
import glob
import os
import sqlite3
import sys


def load(path):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    if len(dbs) > 1:
        sys.exit(f"summarize_prof.py: {len(dbs)} databases under {path} - one traced process per summary "
                 "(bench.py --trace-only-this-config / --no-extras)")
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        mm = {}
        try:
            for n, lo, hi in cur.execute("select name, min(end-start), max(end-start) from kernels group by name"):
                mm[n] = [lo / 1e3, hi / 1e3, float("nan")]
            # bench.py's roofline leg re-launches selected kernels ALONE, 3 + 20 times, right after the timed steps: the last
            # 20 launches of such a kernel are those solo launches (its HIP-event number); the rest ran inside the step
            # next to kernels of other streams
            per = {}
            for n, d in cur.execute("select name, end-start from kernels order by start"):
                per.setdefault(n, []).append(d / 1e3)
            for n, v in per.items():
                if n in mm and len(v) >= 40:
                    mm[n][2] = sum(v[-20:]) / 20.0
        except sqlite3.Error:
            pass
        return dbs[0], [(n, c, t, a, p) + tuple(mm.get(n, (float("nan"),) * 3)) for n, c, t, a, p in rows]
    files = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    return files[0], [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                       float(r["Percentage"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float("nan")) for r in rows]


def main():
    src, rows = load(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    tot = sum(r[2] for r in rows)
    print(f"source: `{os.path.basename(src)}`; total kernel time {tot / 1e3:.2f} ms"
          + (f" over {steps} profiled steps (+warm-up/measurement launches)" if steps else "") + "\n")
    print("`last-20 avg` = mean of the last 20 launches: for the kernels bench.py re-launches alone after the timed steps "
          "(its roofline leg) that is the solo duration its HIP events measure; `avg` mixes those with the in-step launches, "
          "which share the GPU with other streams' kernels.\n")
    print("| kernel | calls | total ms | avg us | min us | max us | last-20 avg us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for n, c, t, a, p, lo, hi, tail in sorted(rows, key=lambda r: -r[2]):
        n = n.split("(")[0] if len(n) > 70 else n
        tl = f"{tail:.2f}" if tail == tail else "—"
        print(f"| `{n}` | {c} | {t / 1e3:.3f} | {a:.2f} | {lo:.2f} | {hi:.2f} | {tl} | {p:.2f} |")


if __name__ == "__main__":
    main()
