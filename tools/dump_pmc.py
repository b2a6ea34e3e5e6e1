"""Raw per-kernel averages of every counter in a rocprofv3 --pmc result directory: python tools/dump_pmc.py DIR [name-substring]"""
import collections, glob, os, sqlite3, sys
res = collections.defaultdict(dict)
for db in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    cur = sqlite3.connect(db).cursor()
    for name, cn, avg, n, dur in cur.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name"):
        k = name.split('(')[0]
        res[k][cn] = avg
        res[k]['dur_us'] = dur / 1000
        res[k]['n'] = n
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, r in sorted(res.items(), key=lambda kv: -kv[1].get('dur_us', 0)):
    if sub not in k:
        continue
    print(k, f"dur {r['dur_us']:.1f} us  n {r['n']}")
    for c, v in sorted(r.items()):
        if c not in ('dur_us', 'n'):
            print(f"    {c:36s} {v:16.1f}")
