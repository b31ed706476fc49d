"""Reads a rocprofv3 kernel-trace CSV and reports, for the busiest stretch of the run, how much of the wall time the kernels cover
and what the gaps between consecutive kernels cost.   python scripts/trace_gaps.py <dir-with-*_kernel_trace.csv> [last_n_kernels]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = rows[-n:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
over = [max(0, rows[i][1] - rows[i + 1][0]) for i in range(len(rows) - 1)]
print(f'{len(rows)} kernels, span {span / 1e6:.3f} ms, sum of durations {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), gaps {sum(gaps) / 1e6:.3f} ms '
      f'(median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us), overlap {sum(over) / 1e6:.3f} ms')
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = k.split('(')[0][:70]
    by[k][0] += 1; by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {t / 1e6:8.3f} ms  n={c:4d}  {t / c / 1e3:7.1f} us  {k}')
