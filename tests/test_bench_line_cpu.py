"""bench.py's LAST stdout line is what the driver parses; it must stay small (BENCH_r05: a 28.8 KB line came back `parsed: null`
and the round went unmeasured).  The assembling function is run here on a full canned record (round 5's, 28.8 KB) and on inflated
ones: the line stays < 4 KB, parses, and carries the contract's keys with `roofline` and `cpu_baseline`."""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'config', 'roofline', 'cpu_baseline')


def _record():
    return json.load(open(os.path.join(REPO, 'tests', 'golden', 'bench_record_r5i.json')))


def _check(line, rec):
    assert len(line) < 4096 and '\n' not in line
    c = json.loads(line)
    for k in CONTRACT:
        assert k in c, k
    assert c['value'] == float('%.6g' % rec['value']) and c['steps'] == rec['steps'] and c['warmup'] == rec['warmup']
    assert abs(c['ms_per_step'] - rec['ms_per_step']) <= 1e-5 * rec['ms_per_step']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms', 'algorithmic_bytes_per_launch'):
        assert k in c['roofline'], k
    assert abs(c['roofline']['frac'] - c['roofline']['achieved'] / c['roofline']['peak']) < 1e-5
    assert 'workload' in c['config'] and 'model' not in c['config'] and c['config']['launch']['mode']
    return c


def test_compact_line_of_a_full_default_record():
    rec = _record()
    assert len(json.dumps(rec)) > 20000                      # the record that did not parse as one line
    c = _check(bench.compact_line(rec, 'bench_detail.json'), rec)
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c['cpu_baseline'], k
    assert c['detail'] == 'bench_detail.json' and set(c['other_ms_per_step']) >= {'c1', 'c2', 'c4'}
    for v in c.values():                                     # no tables / sweeps / legs on the line
        assert not (isinstance(v, dict) and len(json.dumps(v)) > 1500)


def test_compact_line_stays_small_when_strings_grow():
    rec = _record()
    rec['config']['workload'] *= 40
    rec['cpu_baseline']['sample'] *= 40
    rec['config']['launch']['kernels'] = {k: v * 50 for k, v in rec['config']['launch']['kernels'].items()}
    rec['other_workloads'] = {f'w{i}_{"x" * 40}': {'ms_per_step': 1.0 + i} for i in range(120)}
    _check(bench.compact_line(rec, 'd.json'), rec)


def test_compact_line_multi_rank_and_no_cpu_baseline():
    rec = copy.deepcopy(_record())
    rec.update(n_gpus=8, world_size=8, backend='rccl', comm_ms=0.0421, ms_per_step_ranks={'min': 0.39, 'max': 0.41})
    rec['cpu_baseline'] = None                               # N > 1: timed at N = 1 only
    line = bench.compact_line(rec, None)
    c = json.loads(line)
    assert len(line) < 4096 and c['cpu_baseline'] is None and c['comm_ms'] == 0.0421 and c['world_size'] == 8
    assert c['ms_per_step_ranks'] == {'min': 0.39, 'max': 0.41}
    del rec['cpu_baseline']                                  # --no-cpu-baseline: the key is absent, not null
    assert 'cpu_baseline' not in json.loads(bench.compact_line(rec, None))


def test_a_failed_side_workload_does_not_cost_the_line():
    """bench.py records an exception of a multi-rank side workload as {'error': ...} (the headline is what a scaling run is for): the compact
    line carries null for it and stays valid."""
    import bench
    rec = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'bench_record_r5i.json')))
    rec['other_workloads'] = {'strong_c3': {'error': 'RuntimeError: capture refused'}}
    line = bench.compact_line(rec, 'bench_detail.json')
    d = json.loads(line)
    assert len(line) < bench.COMPACT_LIMIT and d['other_ms_per_step'] == {'strong_c3': None} and abs(d['value'] - rec['value']) <= 1e-5 * rec['value']
