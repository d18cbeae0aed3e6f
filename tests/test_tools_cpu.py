"""Host-side checks of what the measurement relies on (no GPU): the CPU-baseline policy of bench.py, the step-timeline
tool and the PMC-traffic summary for stash-mode workloads (launches summed per step)."""
import csv
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_times_the_full_step_when_it_is_affordable(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    from cbfssm import synthetic as syn
    w = syn.tiny(M=12, T=20, B=2, S=4)
    monkeypatch.setenv('CBFSSM_BENCH_THREADS', '2')
    for mode in ('eval', 'train'):
        c = bench.cpu_baseline(w, mode, policy='auto')
        assert c['kind'] == 'port' and c['unit'] == 'steps/s' and c['value'] > 0 and c['cores'] >= 1
        assert c['extrapolated'] is False and c['sample_T'] == w.T and c['sample_gp_calls'] == 3 * w.T - 1
        assert set(c['seconds_per_gp_call_all_cores']) == {'T8', 'T20'}
        assert c['statistic'] == 'median of 5 after 2 warm-ups'          # BASELINE.md section 2, when seven steps are affordable
        assert c['value'] == 1.0 / min(c['threads']['all_cores']['seconds_per_step'],
                                       c['threads']['reference_session_config']['seconds_per_step'])
    # seven full steps do not fit the budget: the median of three
    monkeypatch.setenv('CBFSSM_CPU_BUDGET', '0')
    assert 'median of 3' in bench.cpu_baseline(w, 'eval', policy='auto')['statistic']
    # an unaffordable full step: a T = 128 sample, scaled linearly in the number of GP calls
    monkeypatch.setenv('CBFSSM_CPU_FULL_LIMIT', '0')
    w2 = syn.tiny(M=12, T=200, B=1, S=2)
    c = bench.cpu_baseline(w2, 'eval', policy='auto')
    assert c['extrapolated'] is True and c['sample_T'] == 128
    assert np.isclose(c['threads']['all_cores']['seconds_per_step'] * (3 * 128 - 1) / (3 * 200 - 1),
                      c['seconds_per_gp_call_all_cores']['T128'] * (3 * 128 - 1), rtol=1e-9)


def _trace_csv(path, rows):
    with open(path, 'w', newline='') as f:
        wr = csv.DictWriter(f, fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Grid_Size',
                                           'Workgroup_Size'])
        wr.writeheader()
        for r in rows:
            wr.writerow(dict(zip(wr.fieldnames, r)))


def test_step_timeline_prints_one_step(tmp_path):
    rows = []
    t = 0
    for step in range(4):
        rows.append(('void cbfssm::prepare_kernel<true>(cbfssm::PrepArgs2)', t, t + 100, 1, 2048, 1024))
        rows.append(('void cbfssm::pass_kernel<7, 1, 6, true, 1, 1, false, 3>(cbfssm::PassArgs)', t + 110, t + 900, 2, 1, 448))
        rows.append(('void cbfssm::rev_kernel<7, 1, 6, true, false, 0, 4, true>(cbfssm::RevArgs)', t + 950, t + 4000, 1, 1, 512))
        t += 5000
    p = str(tmp_path / 'trace.csv')
    _trace_csv(p, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'profiles', 'tools', 'step_timeline.py'), p, '1'],
                         capture_output=True, text=True, check=True).stdout
    assert 'step 1: 3 launches, 0.005 ms' in out
    assert 'prepare_kernel' in out and 'rev_kernel' in out and 'queue q0 busy' in out and 'queue q1 busy' in out


def _pmc_csv(path, counter, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w', newline='') as f:
        wr = csv.DictWriter(f, fieldnames=['Kernel_Name', 'Counter_Name', 'Counter_Value'])
        wr.writeheader()
        for name, val in rows:
            wr.writerow({'Kernel_Name': name, 'Counter_Name': counter, 'Counter_Value': val})


def test_traffic_json_sums_the_stash_mode_launches_per_step(tmp_path):
    """Two train-step evaluations (two prepare launches), three adjoint launches and two contractions each: the stash-mode
    kernels are summed per step, the pass kernels keep the largest (= full) launch."""
    rev = 'void cbfssm::rev_kernel<13, 2, 6, false, true, 1, 4, false>(cbfssm::RevArgs)'
    pas = 'void cbfssm::pass_kernel<13, 2, 6, false, 1, 1, false, -1>(cbfssm::PassArgs)'
    con = 'void cbfssm::stash_contract2_kernel<13>(double const*, double const*, long, long, double*)'
    prep = 'void cbfssm::prepare_kernel<false>(cbfssm::PrepArgs2)'
    rows = [(prep, 1.0), (prep, 1.0)] + [(rev, 100.0)] * 6 + [(con, 50.0)] * 4 + [(pas, 10.0), (pas, 40.0), (pas, 40.0)]
    _pmc_csv(str(tmp_path / 'fetch' / 'x' / 'a_counter_collection.csv'), 'FETCH_SIZE', rows)
    _pmc_csv(str(tmp_path / 'write' / 'x' / 'a_counter_collection.csv'), 'WRITE_SIZE', [(n, v / 10.0) for n, v in rows])
    out = str(tmp_path / 'traffic.json')
    subprocess.run([sys.executable, os.path.join(ROOT, 'profiles', 'tools', 'make_traffic_json.py'), 'C4:train',
                    str(tmp_path / 'fetch'), str(tmp_path / 'write'), out], check=True, capture_output=True)
    d = json.load(open(out))['C4:train']
    kib = 1024
    assert d['backward_pass_adjoint'] == int((2.0 * 300.0 + 30.0) * kib)        # 6 launches / 2 steps = 3 per step
    assert d['stash_contraction'] == int((2.0 * 100.0 + 10.0) * kib)
    assert d['backward_pass'] == int((2.0 * 40.0 + 4.0) * kib)                  # the largest dispatch


def test_two_rank_bench_line_carries_its_own_baseline():
    """The N > 1 line of bench.py (the committed two-rank rehearsal on one device): the same workload as N = 1, the
    single-rank step time of that workload measured in the same run, the efficiency that follows, the 8-GPU config (C4)
    with its own baseline, and the collective with its rank count."""
    d = json.loads(open(os.path.join(ROOT, 'profiles', 'r04', 'bench_2ranks_one_device.json')).read().strip().splitlines()[-1])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['workload'].startswith('C3-Sarcos')
    assert set(d['single_rank']) >= {'ms_per_step', 'steps_per_s', 'steps', 'how'}
    assert np.isclose(d['scaling_efficiency'], d['value'] / (d['n_gpus'] * d['single_rank']['steps_per_s']), rtol=1e-9)
    c4 = d['eight_gpu_config']
    assert c4['workload'].startswith('C4-Sarcos-M200') and c4['global_batch'] == 2 * 256
    assert set(c4) >= {'value', 'ms_per_step', 'single_rank', 'scaling_efficiency', 'collective_bytes'}
    assert np.isclose(c4['scaling_efficiency'], c4['value'] / (d['n_gpus'] * c4['single_rank']['steps_per_s']), rtol=1e-9)
    assert d['collective']['ranks'] == 2 and d['collective']['per_step'] == 1 and d['collective']['sum_checked'] is True
