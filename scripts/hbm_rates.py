"""Per-kernel HBM rate = traffic per launch (profiles/<tag>_traffic.json, PMC passes) / average launch duration
(profiles/<tag>_kernel_stats.csv, kernel-trace stats of the same command).    python scripts/hbm_rates.py r02_l > profiles/r02_l_hbm_rates.txt"""
import csv, json, os, sys
tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
traffic = json.load(open(os.path.join(P, f'{tag}_traffic.json')))['kernels']
rows = list(csv.DictReader(open(os.path.join(P, f'{tag}_kernel_stats.csv'))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'HBM rate per kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (profiles/{tag}_traffic.json, separate --pmc passes)')
print(f'/ average launch duration (profiles/{tag}_kernel_stats.csv, rocprofv3 --kernel-trace --stats of the same command).')
print('MI355X HBM3E: 8.0 TB/s spec, ~6.3 TB/s achievable (MI355X_MICROARCH.md).\n')
print('% GPU time launches    avg us  MB/launch   TB/s  kernel')
short = lambda n: n.split('(')[0]
for r in rows[:24]:
    k = short(r['Name'])
    t = next((v for v in traffic if short(v['kernel']) == k), None)
    if t is None:
        continue
    b = t['hbm_bytes_per_launch']
    print(f"{100 * float(r['TotalDurationNs']) / tot:10.2f} {int(r['Calls']):8d} {float(r['AverageNs']) / 1e3:9.1f} {b / 1e6:10.1f} {b / float(r['AverageNs']) / 1e3:6.2f}  {k}")
