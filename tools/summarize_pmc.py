"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as prescribed by
MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of a
wide coalesced read, so reads are doubled; WRITE_SIZE is taken as is (uncalibrated).
Usage: python tools/summarize_pmc.py <fetch_dir> <write_dir> [out.json] [steps] > profiles/xxx_pmc_traffic.md
`steps` = the steps the traced bench command ran (--steps + max(--warmup, 3)): the JSON then carries "_per_step" = the sum
over EVERY dispatch of the trace divided by that count (bench.py's roofline.traffic) and the launches per step per kernel -
derived from the trace, not maintained by hand."""
import glob
import json
import os
import sqlite3
import sys


def per_kernel(d, counter):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg in cur.execute(
            "select name, count(*), avg(counter_value) from pmc_events where counter_name=? group by name", (counter,)):
        out[name.split("(")[0]] = (n, avg)
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        nf, af = f.get(k, (0, 0.0))
        nw, aw = w.get(k, (0, 0.0))
        rd = 2.0 * af * 1024.0          # gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B
        wr = aw * 1024.0
        rows.append((k, max(nf, nw), rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    print("| kernel | launches | HBM read / launch (2 x FETCH_SIZE) | HBM write / launch (WRITE_SIZE) | total MB |")
    print("|---|---:|---:|---:|---:|")
    js = {}
    for k, n, rd, wr in rows:
        if rd + wr < 1e4:
            continue
        print(f"| `{k}` | {n} | {rd / 1e6:.2f} MB | {wr / 1e6:.2f} MB | {(rd + wr) / 1e6:.2f} |")
        js[k] = {"read_bytes": round(rd), "write_bytes": round(wr)}
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    if steps > 0:
        tot = sum(n * (rd + wr) for _, n, rd, wr in rows)
        js["_per_step"] = {"bytes": round(tot / steps), "steps_traced": steps,
                           "launches_per_step": {k: round(n / steps, 2) for k, n, rd, wr in rows if rd + wr >= 1e4}}
        print(f"\nwhole step: {tot / steps / 1e6:.1f} MB of HBM traffic per step (all {sum(r[1] for r in rows)} dispatches of the trace / {steps} steps)")
    out_json = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    with open(out_json, "w") as fh:
        json.dump(js, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
