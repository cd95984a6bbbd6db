"""What bench.py prints: short `{"detail": ...}` lines first, then ONE compact final line (the driver's record).

The final line has exactly the contract's top-level keys, a `config` of at most 25 scalars, a flat `roofline` and a flat `cpu_baseline`; it
is strict JSON (no NaN / Infinity), nested at most two objects deep and shorter than 4 KB.  Everything else measured next to the headline
goes to the detail lines (each a JSON object whose first key is "detail" and which never holds the word "metric") and, unrounded, to
`gpurun_out/bench_detail.json` (or $PDP_BENCH_DETAIL).  tests/test_host_logic.py builds the lines from a committed measurement and checks these properties."""
import json
import math
import os

from . import REPO

TOP_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'collective_backend', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
CONFIG_HEADLINE_KEYS = ('workload', 'E', 'V', 'F', 'iterations_per_step', 'path', 'kernel_launches_per_call', 'kernel_ms_per_launch', 'solve_call_ms',
                        'poison_replay_launches_per_call', 'algorithmic_bytes_per_launch', 'solved_fraction', 'unsat_clauses_total', 'semantics',
                        'parallelism')
# (key of the final line's config, key of benchlib.secondary.driver_summary): the other BASELINE configs, one or two scalars each
CONFIG_SUMMARY_KEYS = (('configs2_it_per_s', 'configs2_np_nd_np_h128_it_per_s'), ('configs2_frac_mfma_f32', 'configs2_np_nd_np_h128_frac_mfma_f32'),
                       ('configs3_shard_it_per_s', 'configs3_shard_n400_it_per_s'), ('configs3_shard_frac_mfma_f32', 'configs3_shard_n400_frac_mfma_f32'),
                       ('configs4_shard_it_per_s', 'configs4_shard_p_nd_np_b4_it_per_s'), ('configs4_shard_frac_mfma_f32', 'configs4_shard_p_nd_np_b4_frac_mfma_f32'),
                       ('train_np_nd_np_frac_mfma_f32', 'train_np_nd_np_frac_mfma_f32'), ('walksat_flips_per_s', 'walksat_1000_flips_per_s'),
                       ('reinforce_it_per_s', 'reinforce_it_per_s'), ('solved_T1000_w1000', 'solved_T1000_w1000_reference_semantics'))
MAX_FINAL_BYTES = 4096
MAX_DETAIL_BYTES = 3500


def _num(v, digits=6):
    "floats to `digits` significant digits; non-finite numbers have no JSON form and become None"
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        r = float('%.*g' % (digits, v))
        return int(r) if r == int(r) and abs(r) < 1e15 and abs(r) >= 10 ** (digits - 1) else r
    try:
        return _num(float(v), digits)                    # numpy scalars
    except (TypeError, ValueError):
        return str(v)


def _clean(obj, digits=6):
    if isinstance(obj, dict):
        return {str(k).replace('metric', 'figure'): _clean(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_clean(v, digits) for v in obj]
    if isinstance(obj, str):
        return obj.replace('metric', 'figure')
    return _num(obj, digits)


def _finite(obj):
    "the measurement as it is, non-finite numbers -> None (strict JSON)"
    if isinstance(obj, dict):
        return {str(k): _finite(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_finite(v) for v in obj]
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if obj is None or isinstance(obj, (bool, int, str)):
        return obj
    try:
        return _finite(float(obj))
    except (TypeError, ValueError):
        return str(obj)


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + '...'


def final_line(full):
    """The compact record from the full measurement dict `full` (what bench.py's main() assembled: the contract keys plus nested `config.secondary`,
    `config.solved`, `config.fast_build`, `roofline.valu_issue`, `cpu_baseline_torch_sparse`, and `summary` = driver_summary(config))."""
    cfg, summ = full.get('config') or {}, full.get('summary') or {}
    config = {k: _num(cfg.get(k)) for k in CONFIG_HEADLINE_KEYS if k in cfg}
    for dst, src in CONFIG_SUMMARY_KEYS:
        if src in summ and isinstance(summ[src], (int, float, str)):
            config[dst] = _num(summ[src])
    assert len(config) <= 25
    rf = full.get('roofline') or {}
    roofline = {k: _num(rf.get(k)) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source')}
    roofline['valu_issue_frac'] = _num((rf.get('valu_issue') or {}).get('issue_frac_at_2_cycles'))
    roofline['kernel'] = rf.get('kernel')
    cpu = full.get('cpu_baseline')
    if cpu is not None:
        ts = full.get('cpu_baseline_torch_sparse') or {}
        cpu = {'value': _num(cpu.get('value')), 'unit': cpu.get('unit'), 'cores': cpu.get('cores'), 'kind': cpu.get('kind'), 'cpu_model': cpu.get('cpu_model'),
               'sample': _short(cpu.get('sample', ''), 200), 'torch_sparse_value': _num(ts.get('value')), 'torch_sparse_cores': ts.get('cores')}
    nested = {'config': config, 'roofline': roofline, 'cpu_baseline': cpu}
    return {k: (nested[k] if k in nested else _num(full.get(k))) for k in TOP_KEYS}


def _split(name, obj, lines):
    "one detail line per object that fits MAX_DETAIL_BYTES; larger dicts are cut along their keys, larger leaves are shortened"
    s = json.dumps({'detail': name, 'data': obj}, allow_nan=False, separators=(',', ':'))
    if len(s) <= MAX_DETAIL_BYTES:
        lines.append(s)
    elif isinstance(obj, dict):
        small = {k: v for k, v in obj.items() if not isinstance(v, (dict, list))}
        small = {k: (_short(v, 300) if isinstance(v, str) else v) for k, v in small.items()}
        if small:
            _split(name, small, lines) if len(json.dumps(small)) + 64 <= MAX_DETAIL_BYTES else lines.append(
                json.dumps({'detail': name, 'data': {k: (_short(v, 60) if isinstance(v, str) else v) for k, v in small.items()}}, allow_nan=False, separators=(',', ':')))
        for k, v in obj.items():
            if isinstance(v, (dict, list)):
                _split(name + '.' + str(k), v, lines)
    elif isinstance(obj, list) and len(obj) > 1:
        h = len(obj) // 2
        _split(name + '[:%d]' % h, obj[:h], lines); _split(name + '[%d:]' % h, obj[h:], lines)
    else:
        lines.append(json.dumps({'detail': name, 'data': _short(s, MAX_DETAIL_BYTES - 200)}, allow_nan=False, separators=(',', ':')))


def detail_lines(full):
    """Everything of `full` the final line does not carry, as short JSON lines: the nested measurements first, the flat per-config summary last
    (closest to the final line: it is what the tail of stdout keeps)."""
    cfg = _clean(full.get('config') or {}, 5)
    lines = []
    for name in ('solved', 'secondary', 'fast_build'):
        if cfg.get(name) is not None:
            _split('config.' + name, cfg[name], lines)
    rf = _clean(full.get('roofline') or {}, 5)
    _split('roofline', {k: rf.get(k) for k in ('valu_issue', 'note') if rf.get(k) is not None}, lines)
    for name in ('cpu_baseline', 'cpu_baseline_torch_sparse'):
        if full.get(name) is not None:
            _split(name, _clean(full[name], 5), lines)
    _split('config.headline', {k: v for k, v in cfg.items() if not isinstance(v, (dict, list)) and k not in (full.get('summary') or {})}, lines)
    _split('summary', _clean(full.get('summary') or {}, 5), lines)
    return lines


def check_final(s):
    "the properties the driver's parser needs; raises AssertionError"
    def bad(c):
        raise ValueError('non-finite constant %s' % c)
    obj = json.loads(s, parse_constant=bad)
    assert '\n' not in s and len(s.encode()) < MAX_FINAL_BYTES, len(s.encode())
    assert tuple(obj.keys()) == TOP_KEYS, list(obj.keys())
    for k, v in obj.items():
        if isinstance(v, dict):
            assert all(not isinstance(x, (dict, list)) for x in v.values()), k
        else:
            assert not isinstance(v, list), k
    assert len(obj['config']) <= 25
    return obj


def emit(full, out=None, side_file=True):
    """Print the detail lines, then the final line (last line of stdout).  The unrounded measurement goes to gpurun_out/bench_detail.json."""
    import sys
    out = out or sys.stdout
    try:
        for l in detail_lines(full):
            assert '"metric"' not in l and len(l) <= MAX_DETAIL_BYTES + 200
            out.write(l + '\n')
    except Exception as ex:                                    # details never cost the record
        out.write(json.dumps({'detail': 'error', 'data': repr(ex)}) + '\n')
    if side_file:
        try:
            path = os.environ.get('PDP_BENCH_DETAIL') or os.path.join(REPO, 'gpurun_out', 'bench_detail.json')
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as f:
                json.dump(_finite(full), f, allow_nan=False)
        except Exception:
            pass
    s = json.dumps(final_line(full), allow_nan=False, separators=(', ', ': '))
    check_final(s)
    out.write(s + '\n')
    out.flush()
    return s
