"""One step's kernel timeline from a rocprofv3 --kernel-trace run (rocpd .db): start offset, duration, queue.
Usage: python tools/timeline.py <dir with the .db> [step index from the end, default 2]"""
import glob
import os
import sqlite3
import sys


def main():
    db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
    # a step ends with its Adam + EMA kernel (the step state is advanced inside the loss / heads-backward kernel)
    adv = [i for i, r in enumerate(rows) if "k_adam_ema" in r[0]]
    a, b = adv[-back - 1] + 1, adv[-back] + 1
    t0 = rows[a][1]
    print(f"step of {b - a} kernels, {(rows[b - 1][2] - t0) / 1e3:.1f} us")
    busy_end = t0
    for name, s, e, q, st in rows[a:b]:
        gap = (s - busy_end) / 1e3
        busy_end = max(busy_end, e)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {'idle %.1f' % gap if gap > 1.0 else '':10s} {name.split('(')[0][:60]}")


if __name__ == "__main__":
    main()
