"""Per-kernel MFMA-pipe utilisation and wait fractions from tools/pmc_conv.sh's rocprofv3 --pmc run (SQ counters on
solo kernel replays): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 8 ... ) normalised as in MI355X_MICROARCH.md
(busy cycles / 32 / GRBM_GUI_ACTIVE).  Usage: python tools/show_pmc.py [--md] > profiles/xxx_pmc_mfma_busy.md"""
import sqlite3, collections, glob, sys
res = collections.defaultdict(dict)
pats = [a for a in sys.argv[1:] if not a.startswith('--')] or ['/root/repo/gpurun_out/pmcq_*/r01_results.db']
for db in sorted(set(d for p in pats for d in glob.glob(p, recursive=True))):
    cur = sqlite3.connect(db).cursor()
    for name, cn, avg, n, dur in cur.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name"):
        k = name.split('(')[0]
        res[k][cn] = avg
        res[k]['dur_us'] = dur / 1000
md = '--md' in sys.argv
if md:
    print('| kernel (solo replay) | us | clock GHz | MFMA pipe busy | wave cycles waiting (any) | waiting on instruction issue | LDS bank-conflict cycles (summed over all CUs) |')
    print('|---|---:|---:|---:|---:|---:|---:|')
for k, r in sorted(res.items(), key=lambda kv: -kv[1].get('dur_us', 0)):
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in r or r['dur_us'] < 10:
        continue
    clk = r.get('GRBM_GUI_ACTIVE', 0) / r['dur_us'] / 1e3
    mf = r['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 / max(r.get('GRBM_GUI_ACTIVE', 1), 1)
    # GRBM_GUI_ACTIVE also counts cycles outside the dispatch (launch, drain): for kernels under ~20 us the "clock" comes out
    # above the part's 2.4 GHz and every busy / wait FRACTION of such a row is diluted by the same factor - the column is
    # printed only where it means something, short kernels are marked
    short = r['dur_us'] < 20.0
    if md:
        clk_s = "n/a (< 20 us: GRBM_GUI_ACTIVE includes cycles outside the dispatch; fractions in this row are diluted)" if short else f"{clk:.2f}"
        print(f"| `{k}` | {r['dur_us']:.1f} | {clk_s} | {mf:.3f} | {r.get('SQ_WAIT_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f} | {r.get('SQ_WAIT_INST_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f} | {r.get('SQ_LDS_BANK_CONFLICT',0):.0f} |")
        continue
    print(f"{k:40s} {r['dur_us']:8.1f} us  clk {clk:.2f} GHz  mfma_util {mf:.3f}  wait_any/wave {r.get('SQ_WAIT_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f}  wait_inst/wave {r.get('SQ_WAIT_INST_ANY',0)/max(r.get('SQ_WAVE_CYCLES',1),1):.2f}  bankconf {r.get('SQ_LDS_BANK_CONFLICT',0):.0f}")
