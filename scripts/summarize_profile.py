"""Turn the rocprofv3 outputs merged into gpurun_out/ into the summaries committed under profiles/.
   python scripts/summarize_profile.py TAG STATS_DIR FETCH_DIR WRITE_DIR BENCH_JSON"""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
tag, stats_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
out = 'profiles'
shutil.copy(glob.glob(f'{stats_dir}/*/*kernel_stats.csv')[0], f'{out}/{tag}_kernel_stats.csv')
shutil.copy(bench_json, f'{out}/{tag}_bench.json')


def per_kernel(d, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(f'{d}/*/*counter_collection.csv')[0])):
        if r['Counter_Name'] == counter:
            k = r['Kernel_Name'].split('(')[0]
            tot[k][0] += 1
            tot[k][1] += float(r['Counter_Value'])
    return tot


F, W = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
rows = []
for k in sorted(set(F) | set(W), key=lambda k: -(F.get(k, [0, 0])[1] + W.get(k, [0, 0])[1])):
    nf, f = F.get(k, [0, 0.0]); nw, w = W.get(k, [0, 0.0])
    n = max(nf, nw)
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B
    # for wide coalesced reads -> x2; WRITE_SIZE is exact for 16-B-per-lane stores
    rows.append(dict(kernel=k, launches=n, fetch_kib_raw_per_launch=round(f / max(nf, 1), 1), write_kib_per_launch=round(w / max(nw, 1), 1),
                     hbm_bytes_per_launch=int((2 * f / max(nf, 1) + w / max(nw, 1)) * 1024)))
dom = [r for r in rows if r['kernel'].startswith('void k_conv_mfma<3')]
n = sum(r['launches'] for r in dom)
traffic = sum(r['hbm_bytes_per_launch'] * r['launches'] for r in dom) / max(n, 1)
json.dump(dict(tag=tag, csrc_digest=csrc_digest(), note='HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes (MI355X_MICROARCH.md, HBM section)',
               dominant_kernel='k_conv_mfma<3,...> (all template variants)', dominant_launches=n, dominant_hbm_bytes_per_launch=int(traffic),
               kernels=rows[:25]), open(f'{out}/{tag}_traffic.json', 'w'), indent=1)
print(f'{tag}: dominant kernel {n} launches, {traffic / 1e6:.1f} MB per launch')
