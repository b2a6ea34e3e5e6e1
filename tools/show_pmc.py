import sqlite3, collections, glob, sys
res = collections.defaultdict(dict)
for db in glob.glob('/root/repo/gpurun_out/pmcq_*/r01_results.db'):
    cur = sqlite3.connect(db).cursor()
    for name, cn, avg, n, dur in cur.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name"):
        k = name.split('(')[0]
        res[k][cn] = avg
        res[k]['dur_us'] = dur / 1000
for k, r in res.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in r or r['dur_us'] < 10:
        continue
    clk = r.get('GRBM_GUI_ACTIVE', 0) / r['dur_us'] / 1e3
    mf = r['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 / max(r.get('GRBM_GUI_ACTIVE', 1), 1)
    print(f"{k:40s} {r['dur_us']:8.1f} us  clk {clk:.2f} GHz  mfma_util {mf:.3f}  wait_any/wave {r.get('SQ_WAIT_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f}  wait_inst/wave {r.get('SQ_WAIT_INST_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f}  bankconf {r.get('SQ_LDS_BANK_CONFLICT',0):.0f}")
