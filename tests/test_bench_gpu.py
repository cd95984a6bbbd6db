"""bench.py on the GPU box: the line's contract fields at N = 1 (with the secondary measurements and both CPU baselines on a small
workload) and the launcher at N = 2 (two gloo ranks sharing the box's single GPU: RCCL wants one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

from helpers import REPO

pytestmark = pytest.mark.gpu


def _strict(s):
    def bad(c):
        raise ValueError(c)
    return json.loads(s, parse_constant=bad)


def _bench(env_extra, *argv, detail=None):
    """(record, full): the record is the LAST stdout line (compact, < 4 KB, the only line with "metric"); `full` is the unrounded
    measurement the run wrote next to it (None without `detail`)."""
    env = dict(os.environ); env.update(env_extra)
    if detail is not None:
        env['PDP_BENCH_DETAIL'] = str(detail)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.split('\n') if l.strip()]
    assert lines[-1].startswith('{"metric"') and len(lines[-1].encode()) < 4096
    assert sum('"metric"' in l for l in lines) == 1
    for l in lines[:-1]:
        if l.startswith('{'):
            assert list(_strict(l).keys()) == ['detail', 'data'] and len(l) < 4096
    rec = _strict(lines[-1])
    for k, v in rec.items():
        if isinstance(v, dict):
            assert all(not isinstance(x, (dict, list)) for x in v.values()), k
    return rec, (json.load(open(str(detail))) if detail is not None else None)


def test_bench_line_small_workload(tmp_path):
    rec, line = _bench({}, '--steps', '2', '--warmup', '1', '--batch', '600', '--iters', '40', '--cpu-cores', '8', '--secondary-walksat-steps', '200',
                       '--config3-batch', '200', '--config4-instances', '30', detail=tmp_path / 'detail.json')
    assert tuple(rec.keys()) == ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'collective_backend', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
    assert len(rec['config']) <= 25 and rec['config']['path'] == 'persistent-lds' and rec['config']['iterations_per_step'] == 40
    assert abs(rec['value'] - line['value']) < 1e-5 * line['value'] and rec['roofline']['kernel'] == line['roofline']['kernel']
    assert abs(rec['roofline']['frac'] - rec['roofline']['achieved'] / rec['roofline']['peak']) < 1e-5 and rec['roofline']['bound'] == 'hbm'
    assert rec['cpu_baseline']['kind'] == 'port' and rec['cpu_baseline']['cores'] == 8 and rec['cpu_baseline']['value'] > 0 and rec['cpu_baseline']['torch_sparse_value'] > 0
    for k in ('configs2_it_per_s', 'configs2_frac_mfma_f32', 'configs3_shard_frac_mfma_f32', 'configs4_shard_frac_mfma_f32', 'train_np_nd_np_frac_mfma_f32',
              'walksat_flips_per_s', 'reinforce_it_per_s'):
        assert rec['config'][k] > 0, k
    for k in ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline', 'cpu_baseline_torch_sparse'):
        assert k in line, k
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1 and line['steps'] == 2 and line['vs_baseline'] is None and line['dtype'] == 'f32'
    assert line['config']['path'] == 'persistent-lds' and line['config']['iterations_per_step'] == 40
    rf = line['roofline']
    assert rf['bound'] == 'hbm' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and rf['peak'] == 8000.0
    assert abs(rf['achieved'] - line['config']['algorithmic_bytes_per_launch'] / (line['config']['kernel_ms_per_launch'] * 1e-3) / 1e9) < 1e-6 * rf['achieved']
    cpu = line['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] == 8 and cpu['value'] > 0 and 'cpu_model' in cpu
    ts = line['cpu_baseline_torch_sparse']
    assert ts['runs'] and ts['runs'][0]['B'] == 500 and ts['runs'][0]['seconds_per_iteration'] > 0
    sec = line['config']['secondary']
    for name in ('neural', 'walksat', 'reinforce'):
        assert 'error' not in sec[name], sec[name]
    nk = sec['neural']['kernels']
    for key in ('agg_pre', 'agg_post', 'gru', 'predict_head', 'aggregator_call'):
        assert nk[key].get('ms_per_launch', nk[key].get('ms')) > 0 and 0 < nk[key]['frac_of_mfma_f32_peak'] < 1.0
    assert nk['gru']['launches'] == 2 * sec['neural']['iterations'] and nk['agg_post']['launches'] == 2 * sec['neural']['iterations']
    assert nk['gru']['kernel'].startswith('k_gru_pipe<65') and nk['agg_pre']['kernel'].startswith('k_agg_pre_wave<65') and nk['agg_post']['kernel'].startswith('k_agg_post_pf<')
    assert sec['walksat']['steps'] > 0 and sec['walksat']['kernel_launches'] >= 1 and sec['walksat']['kernel'].startswith('k_walksat<') and 'roofline' not in sec['walksat']
    assert sec['walksat']['flips_per_sec'] > 0
    assert rf['kernel'].startswith('k_sp_solve_lds<false, false, false') and sec['reinforce']['kernel'].startswith('k_sp_solve_lds<true, false, true')
    # every BASELINE config has a driver-visible entry: configs[3] / configs[4] at the per-GPU shape (small sizes here)
    c3, c4 = sec['config3_shard'], sec['config4_shard']
    for c in (c3, c4):
        assert 'error' not in c, c
        assert c['seconds'] > 0 and 0 < c['roofline']['frac'] < 1.0 and c['walksat']['steps'] > 0 and c['kernels']['gru']['ms_per_launch'] > 0
        assert abs(c['roofline']['achieved'] - c['flop_total'] / c['seconds'] / 1e12) < 1e-9 * c['roofline']['achieved']
    assert c3['model_type'] == 'np-nd-np' and c3['instances'] == 200 and c3['segments'] == [200]
    assert c4['model_type'] == 'p-nd-np' and c4['batch_replication'] == 4 and sum(c4['segments']) == 30 and c4['kernels']['gru']['kernel'].startswith('k_gru_pipe<2')
    assert sec['reinforce']['path'] == 'persistent-lds' and sec['reinforce']['kernel_ms_per_launch'] > 0
    for name, row in line['config']['solved'].items():
        assert 'error' not in row, row
        assert 0 <= row['solved'] <= row['instances'] == 600
    # the opt-in fast build is measured next to the headline, never instead of it: same kernel, same workload, its own library
    fb = line['config']['fast_build']
    assert 'error' not in fb, fb
    assert fb['kernel'] == rf['kernel'] and fb['value'] > 0 and fb['kernel_ms_per_launch'] > 0 and 0 < fb['roofline_frac'] < 2.0
    assert 'error' not in fb['neural'] and fb['neural']['kernels']['agg_post']['ms_per_launch'] > 0
    assert abs(fb['solved_fraction'] - fb['solved_fraction_parity_build']) <= 0.01
    # what the driver's record keeps (scalars of `config`, the tail of stdout): every BASELINE config's figures, flat, and last in the line
    sm = line['summary']
    for key in ('configs2_np_nd_np_h128_it_per_s', 'configs2_np_nd_np_h128_frac_mfma_f32', 'configs3_shard_n400_frac_mfma_f32', 'configs4_shard_p_nd_np_b4_frac_mfma_f32',
                'configs2_kernel_agg_post_frac_mfma_f32', 'configs2_kernel_gru_ms', 'fast_build_it_per_s', 'fast_build_configs2_it_per_s', 'walksat_1000_flips_per_s',
                'reinforce_it_per_s', 'train_np_nd_np_frac_mfma_f32'):
        assert isinstance(sm[key], float) and sm[key] > 0, key
    assert 'fast_build_configs2_frac_mfma_f32' not in sm          # bf16x3 products are never priced against the fp32 MFMA peak
    assert abs(sm['configs2_np_nd_np_h128_frac_mfma_f32'] - sec['neural']['roofline']['frac']) < 1e-9 and abs(rec['config']['configs2_frac_mfma_f32'] - sm['configs2_np_nd_np_h128_frac_mfma_f32']) < 1e-5
    assert abs(sm['configs3_shard_n400_frac_mfma_f32'] - c3['roofline']['frac']) < 1e-9 and abs(sm['fast_build_it_per_s'] - fb['value']) < 1e-6 * fb['value']


def test_bench_launcher_two_ranks_on_one_gpu():
    line, _ = _bench({'PDP_DIST_BACKEND': 'gloo'}, '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '500', '--iters', '30')
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['scaling'] == 'weak' and line['collective_backend'] == 'gloo'
    assert line['cpu_baseline'] is None and 'configs2_it_per_s' not in line['config']
    one, _ = _bench({}, '--steps', '2', '--warmup', '1', '--batch', '500', '--iters', '30', '--no-cpu-baseline', '--no-secondary', '--no-fast-build')
    assert one['n_gpus'] == 1 and one['config']['E'] == line['config']['E']          # per-rank batch is fixed: weak scaling
    assert line['config']['iterations_per_step'] == one['config']['iterations_per_step']
