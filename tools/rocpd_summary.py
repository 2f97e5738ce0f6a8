"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel table, stamped with the library's build id.
Usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from mcgaze_amd import lib as _L
    STAMP = f'library build id {_L.build_id()} (sources in tree: {_L.source_id()})'
except Exception as e:   # the summary is still worth having
    STAMP = f'library build id unknown ({e})'
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc').fetchall()
total = sum(r[2] for r in rows)
out = [STAMP, '', '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
for n, c, s, a, lo, hi in rows:
    out.append(f'| `{n[:110]}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {lo / 1e3:.1f} | {hi / 1e3:.1f} | {100 * s / total:.2f} |')
text = '\n'.join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text + '\n')
print(text)
