"""bench.py as its own multi-rank launcher (SURVEY.md section 8 rows d/e; reference analogue engine/train.py:1454-1470):
`python bench.py --gpus N` must start N ranks itself before any GPU call, wire RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*,
relay rank 0's line with n_gpus == N, and fail when a rank fails.  Exercised on CPU over gloo with the stand-in timed
region (MUDIFF_BENCH_DRYRUN=1); the real worker shares the rank plumbing."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, 'bench.py')


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(MUDIFF_BENCH_DRYRUN='1', **env)
    return subprocess.run([sys.executable, BENCH, *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def _line(p):
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    p = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p)
    assert d['n_gpus'] == 2 and d['ranks_seen'] == [0, 1] and d['scaling'] == 'weak'
    assert d['ms_per_step'] >= 20.0          # MAX over ranks: rank 1's stand-in region is the longer one (2 x 10 ms)
    # N-rank reporting (north_star: "RCCL broadcast of params ... at load only"; reference engine/train.py:188-190): every rank's
    # own rate next to the MAX-over-ranks headline, and the load-time broadcast in bytes / ms (the dry run broadcasts two
    # small Linear modules over gloo and asserts that rank 0's parameters arrived everywhere)
    pr, pb = d['per_rank'], d['param_broadcast']
    assert len(pr['slices_per_s']) == 2 and pr['min'] <= pr['max'] and pr['slices_per_s'][1] < pr['slices_per_s'][0]
    assert pb['bytes'] == 4 * (8 * 4 + 4 + 4 * 2 + 2) and pb['messages'] == 2 and pb['ms'] > 0


def test_strong_scaling_shards_cover_the_total_once():
    p = _run(['--gpus', '3', '--total-slices', '37', '--batch', '4'])
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p)
    assert d['n_gpus'] == 3 and d['scaling'] == 'strong'
    assert d['shards'] == [[0, 13], [13, 25], [25, 37]]


def test_a_failing_rank_fails_the_launcher_and_stops_the_others():
    p = _run(['--gpus', '2'], MUDIFF_BENCH_FAIL_RANK='1')       # rank 0 would wait for rank 1 in the rendezvous forever
    assert p.returncode != 0
    assert 'rank 1 exited with code 3' in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def test_world_size_must_equal_gpus_under_an_external_launcher():
    p = _run(['--gpus', '8'], WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)


def test_single_rank_needs_no_launcher():
    d = _line(_run(['--gpus', '1']))
    assert d['n_gpus'] == 1 and d['ranks_seen'] == [0]


def test_sweep_prints_one_line_with_the_counts_that_fit():
    p = _run(['--sweep', '1,2,4', '--total-slices', '24', '--batch', '4'], MUDIFF_BENCH_VISIBLE_GPUS='2')
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p)
    assert sorted(d['slices_per_s_by_gpus']) == ['1', '2'] and d['skipped_gpu_counts'] == [4] and d['scaling'] == 'strong'
    assert d['ranks_seen_by_gpus'] == {'1': [0], '2': [0, 1]} and d['total_slices'] == 24
    assert set(d['efficiency_vs_1gpu']) == {'1', '2'} and d['efficiency_vs_1gpu']['1'] == 1.0
    assert list(d['per_rank_by_gpus']) == ['2'] and d['param_broadcast_by_gpus']['2']['messages'] == 2


def test_sweep_whose_first_count_does_not_fit_still_carries_the_extras():
    """--sweep 8,2,1 on a 2-GPU box used to pass --no-cpu-baseline / --no-roofline to every run that executed
    (they rode on counts[0] only); now they ride on the first count that actually runs."""
    sys.path.insert(0, REPO)
    import bench
    calls = []

    def fake_launch(n, args, **kw):
        calls.append((n, list(args)))
        return 0, json.dumps({'value': 10.0 * n, 'ms_per_step': 1.0, 'ranks_seen': list(range(n)), 'cpu_baseline': {'value': 0.1} if '--no-cpu-baseline' not in args else None,
                              'roofline': {'frac': 0.1} if '--no-roofline' not in args else None})
    old = bench.launch_ranks, bench.visible_gpus
    bench.launch_ranks, bench.visible_gpus = fake_launch, (lambda: 4)
    try:
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = bench.run_sweep(bench.parse(['--sweep', '8,4,2']), ['--sweep', '8,4,2'])
    finally:
        bench.launch_ranks, bench.visible_gpus = old
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert rc == 0 and d['skipped_gpu_counts'] == [8] and [c[0] for c in calls] == [4, 2]
    assert '--no-cpu-baseline' not in calls[0][1] and '--no-roofline' not in calls[0][1]
    assert '--no-cpu-baseline' in calls[1][1] and '--no-roofline' in calls[1][1]
    assert d['cpu_baseline'] == {'value': 0.1} and d['roofline'] == {'frac': 0.1} and d['efficiency_vs_1gpu'] is None


def test_launcher_helpers():
    sys.path.insert(0, REPO)
    import bench
    assert bench.last_json_line('noise\n{"a": 1}\ntrailing') == {'a': 1}
    assert bench.last_json_line('nothing here') is None
    assert len(bench.csrc_digest()) == 16
    a = bench.parse(['--gpus', '4', '--total-slices', '512'])
    assert a.gpus == 4 and a.total_slices == 512 and a.batch == 32      # default batch = BASELINE config 3


@pytest.mark.parametrize('which', ['g1', 'g2'])
def test_seeded_state_dict_equals_the_fixture_weight_table(which):
    """mudiff_hip.weights.seeded_state_dict (used by bench.py's parity leg) must reproduce the weights the fixtures were
    made with (oracle.make_state_dict), tensor for tensor."""
    import torch
    from oracle import mudiff_oracle as O
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    from mudiff_hip.weights import seeded_state_dict
    cfg = O.default_config(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(16,), num_res_blocks=1, embedding_type='fourier')
    m = (NCSNpp if which == 'g1' else NCSNpp_adaptive)(cfg)
    ours, ref = seeded_state_dict(m, which, 1234, cfg.fourier_scale), O.make_state_dict(cfg, which, 1234)
    assert list(ours) == list(ref)
    assert all(torch.equal(ours[k], ref[k]) for k in ref)


@pytest.mark.gpu
def test_two_real_ranks_on_the_gpu_box():
    """The REAL multi-rank worker (models, parameter broadcast, hipGraph capture next to a live process group, barrier, MAX over
    ranks, rank-0 line) with two ranks sharing cuda:0 over gloo - every line of the N-rank path except RCCL itself, which needs
    one GPU per rank.  Weak (2 x 2 slices) and strong (5 slices over 2 ranks, ragged tail batch)."""
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'MUDIFF_BENCH_DRYRUN')}
    e.update(MUDIFF_BENCH_BACKEND='gloo', MUDIFF_BENCH_SAME_GPU='1')
    common = ['--gpus', '2', '--batch', '2', '--steps', '1', '--warmup', '1', '--no-extras', '--no-cpu-baseline', '--no-roofline']
    for extra, scaling in (([], 'weak'), (['--total-slices', '5'], 'strong')):
        p = subprocess.run([sys.executable, BENCH, *common, *extra], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        d = _line(p)
        assert d['n_gpus'] == 2 and d['ranks_seen'] == [0, 1] and d['scaling'] == scaling and d['value'] > 0
        assert d['config']['hipgraph'] is True
        # the generators' parameters: 20,472,065 (G1) + 21,399,681 (G2) fp32 values in two flattened messages
        assert d['param_broadcast']['bytes'] == 4 * (20472065 + 21399681) and d['param_broadcast']['messages'] == 2
        assert len(d['per_rank']['slices_per_s']) == 2 and 0 < d['per_rank']['min'] <= d['per_rank']['max']
