"""Summarise a rocprofv3 --pmc run (rocpd sqlite output): per (kernel, grid) mean of each counter and the duration."""
import glob, sqlite3, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_mfma'
for fn in glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True):
    db = sqlite3.connect(fn)
    q = '''select k.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.end - d.start, p.name, e.value,
                  k.arch_vgpr_count, k.accum_vgpr_count, d.group_segment_size, d.dispatch_id
           from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_info_kernel_symbol k on d.kernel_id = k.id'''
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, gx, gy, wx, dur, cname, val, vg, ag, lds, did in db.execute(q):
        if pat not in name:
            continue
        key = (name.split('(')[0][-70:], gx // wx, gy, wx, vg, ag, lds)
        rows[key][cname].append(val)
        rows[key]['_dur_us'].append(dur / 1e3)
    for key, cs in rows.items():
        print('kernel ..%s  blocks %d x %d  threads %d  vgpr %d agpr %d lds %d' % key)
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0)
        for k, v in sorted(m.items()):
            print(f'    {k:28s} {v:16.1f}' + (f'  {v / wc * 100:6.1f}% of WAVE_CYCLES' if wc and k.startswith('SQ_') else ''))
