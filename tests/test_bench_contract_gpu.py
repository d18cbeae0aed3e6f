"""The driver's bench contract: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the agreed
keys, including the `roofline` and `cpu_baseline` objects (run on the smallest workload so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True,
                         timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_bench_line_has_the_contract_keys(mode):
    rec = _run('--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'C1', '--mode', mode)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in rec, k
    assert rec['n_gpus'] == 1 and rec['steps'] == 3 and rec['warmup'] == 1
    assert rec['higher_is_better'] is True and rec['scaling'] == 'weak' and rec['vs_baseline'] is None
    assert rec['dtype'] == 'f64' and rec['data'] == 'synthetic' and 'workload' in rec['config']
    assert rec['value'] == pytest.approx(1e3 / rec['ms_per_step'], rel=1e-6)
    roof = rec['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'hbm_kernel'):
        assert k in roof, k
    assert roof['bound'] == 'mfma' and roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert roof['hbm_kernel']['bound'] == 'hbm' and roof['hbm_kernel']['achieved'] > 0
    cpu = rec['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cpu, k
    assert cpu['kind'] == 'port' and cpu['value'] > 0 and cpu['cores'] >= 1


def test_float32_bench_line_has_a_roof():
    """`--dtype float32` (the reference's model dtype argument): the float32 kernels timed like the float64 ones, against the
    float32 matrix peak."""
    rec = _run('--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'C1', '--mode', 'train', '--dtype', 'float32',
               '--no-cpu-baseline')
    roof = rec['roofline']
    assert rec['dtype'] == 'f32' and roof['bound'] == 'mfma' and roof['peak'] == pytest.approx(157.3)
    assert roof['frac'] == pytest.approx(roof['achieved'] / roof['peak']) and roof['achieved'] > 0
    assert set(roof['kernel_ms']) == {'backward_pass_f32', 'forward_pass_f32', 'forward_pass_adjoint_f32',
                                      'backward_pass_adjoint_f32'}


@pytest.mark.parametrize('model', ['half', 'prssm'])
def test_variant_bench_lines(model):
    """`--model half|prssm`: the forward-only variants have a bench line with a roof."""
    rec = _run('--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'C1', '--model', model)
    assert rec['config']['model'] == model and rec['config']['mode'] == 'train' and rec['value'] > 0
    roof = rec['roofline']
    assert roof['bound'] == 'mfma' and roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert set(roof['kernel_ms']) == {'forward_pass', 'forward_pass_adjoint'}
    # ... and in float32 arithmetic against the float32 matrix peak
    rec = _run('--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'C1', '--model', model, '--dtype', 'float32')
    assert rec['dtype'] == 'f32' and rec['roofline']['peak'] == pytest.approx(157.3) and rec['value'] > 0
    assert set(rec['roofline']['kernel_ms']) == {'forward_pass', 'forward_pass_adjoint'}


def test_two_rank_bench_line_on_one_device():
    """The N > 1 path of bench.py end to end, two ranks sharing this one GPU over gloo (CBFSSM_BENCH_ONE_DEVICE=1; the real run
    is one rank per GPU over RCCL): the default workload is the N = 1 one, and the line carries its own single-rank baseline, the
    efficiency that follows from it and the 8-GPU config (C4) with its own baseline."""
    env = dict(os.environ, CBFSSM_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3',
                          '--warmup', '1'], capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['workload'].startswith('C3-Sarcos') and rec['config']['global_batch'] == 512
    assert rec['single_rank']['steps_per_s'] > 0
    assert rec['scaling_efficiency'] == pytest.approx(rec['value'] / (2 * rec['single_rank']['steps_per_s']), rel=1e-9)
    c4 = rec['eight_gpu_config']
    assert c4['workload'].startswith('C4-Sarcos-M200') and c4['single_rank']['steps_per_s'] > 0
    assert rec['collective']['ranks'] == 2 and rec['collective']['sum_checked'] is True
