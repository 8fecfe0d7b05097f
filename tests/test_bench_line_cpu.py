"""The bench line's contract, checked on CPU against the line the final GPU pass committed (profiles/r05_zzz_bench_default.json): the keys the
driver reads, the two objects the tier asks for (`roofline`, `cpu_baseline`), and the compact `summary` -- the LAST key, small enough that a
record keeping only the tail of the line still has every workload's numbers -- rebuilt here by bench.summary_rows from the line's own fields."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LINE = os.path.join(REPO, 'profiles', 'r05_zzz_bench_default.json')


def _line():
    rows = [ln for ln in open(LINE).read().splitlines() if ln.strip()]
    assert len(rows) == 1, 'bench.py prints ONE line'
    return json.loads(rows[0])


def test_contract_keys_and_objects():
    d = _line()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'node-frames/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf, cb = d['roofline'], d['cpu_baseline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    assert cb['kind'] in ('reference', 'port') and cb['cores'] >= 1 and cb['value'] > 0 and cb['sample']
    # value = the node-frames all ranks processed / the timed seconds
    c = d['config']
    units = c['rooms_per_gpu'] * d['n_gpus'] * c['nodes'] * c['frames']
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 / units - 1) < 1e-3


def test_summary_is_last_small_and_faithful():
    import bench
    d = _line()
    assert list(d)[-1] == 'summary'
    s = d['summary']
    assert len(json.dumps(s)) < 1500
    names = [k for k in s if k != '_cols']
    assert set(d['configs']) <= set(names) and len(names) == len(d['configs']) + 1
    head_name = [k for k in names if k not in d['configs']][0]
    rebuilt = bench.summary_rows(head_name, d, d.get('parity_sample'), d['configs'])
    assert rebuilt == s
    # every workload row: ms, x real-time, pipeline fraction, kernel, its fraction, traffic ratio, parity
    for nm in names:
        row = s[nm]
        src = d if nm == head_name else d['configs'][nm]
        assert row[0] == round(src['ms_per_step'], 3) and row[1] == round(src['x_realtime'], 1)
        assert row[6] is not None and row[6] < 1e-4
    assert 'stream_x_realtime_by_hops' in s['online1'][7] and 'hipgraph' in s['C2'][7]


def test_ill_posed_instances_are_classified_from_the_masks_and_reported():
    """bench.py ILL_POSED_WEIGHT: a room whose PREDICTED masks leave a statistic of some bin without frames is reported, not asserted."""
    import numpy as np
    import bench
    rng = np.random.default_rng(0)
    good = [rng.uniform(0.05, 0.95, size=(5, 40)) for _ in range(2)]
    assert bench.min_statistic_weight((good, good)) > 1e-2
    bad = [m.copy() for m in good]
    bad[1][3, :] = 1.0 - 1e-6                                  # one bin saturated in every frame: sum_t (1 - m)^2 = 40e-12
    w = bench.min_statistic_weight((good, bad))
    assert w < bench.ILL_POSED_WEIGHT and abs(w - 40e-12) < 1e-12
    # the committed line: C4 samples 32 rooms, the ill-posed ones are listed with error and weight and do not enter worst_rel
    ps = _line()['configs']['C4']['parity_sample']
    ip = ps['ill_posed']['rooms']
    assert len(ps['per_room']) + len(ip) == 32 and ps['ok'] and ps['worst_rel'] < 1e-4
    assert all(v['min_statistic_weight'] < bench.ILL_POSED_WEIGHT for v in ip.values())
    assert not set(ip) & set(ps['per_room'])
    row = _line()['summary']['C4']
    assert row[-1]['ill_posed_rooms'] == [len(ip), 32]
