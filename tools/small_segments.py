#!/usr/bin/env python3
"""The small-segment regime of the neural path (round-4 verdict, item 5): with the reference's DEFAULT memory limit `-l 4e7`
(satyr.py:53; dataset.py:36,57 divide it by the hidden dimension) configs[2]'s 5 000 instances fall into 41 segments of 124 instances
(0.31 M edges each), configs[4]'s mixed batch with -b 4 into many more -- one forward per segment, >= 8 launches and one host read per
sweep.  Reports, next to the one-segment numbers (`-l 4e9`): wall time, instance-sweeps per second, the sum of the library's HIP-event kernel
times and the GPU-busy fraction (kernel time / wall time).

    python tools/small_segments.py [--instances 5000] [--iters 10] [--config4-instances 600]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
sys.path.insert(0, os.path.join(REPO, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--instances', type=int, default=5000)
    ap.add_argument('--n', type=int, default=200)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--config4-instances', type=int, default=600)
    ap.add_argument('--limits', default='40000000,4000000000')
    args = ap.parse_args()
    args.batch, args.tolerance, args.t_max = args.instances, 0.02, 100
    import torch
    from pdp import native
    from pdp.factorgraph import dataset
    from benchlib.neural import neural_shard, config4_items
    native.require_gpu()
    dev = torch.device('cuda:0')
    rows = []
    items2 = dataset.random_ksat_items(args.instances, args.n, 3, m=int(round(4.2 * args.n)), seed=0)
    items4 = config4_items(args.config4_instances)
    for name, items, mt, rep in (("configs[2] np-nd-np", items2, 'np-nd-np', 1), ("configs[4] p-nd-np -b 4", items4, 'p-nd-np', 4)):
        for limit in [int(x) for x in args.limits.split(',')]:
            r = neural_shard(args, dev, native, items, mt, args.hidden, args.iters, replication=rep, limit=limit, workload=name)
            kms = sum(v['ms_per_launch'] * v['launches'] for k, v in r['kernels'].items() if 'launches' in v)
            sweeps = sum(r['iterations_per_segment'])
            inst_sweeps = sum(n * it for n, it in zip(r['segments'], r['iterations_per_segment']))
            row = dict(workload=name, limit=limit, segments=len(r['segments']), instances_per_segment_max=max(r['segments']), sweeps=sweeps,
                       wall_s=r['seconds'], loop=r['path'], wall_s_with_kernel_events=r.get('seconds_with_kernel_events'), instance_sweeps_per_s=inst_sweeps * rep / r['seconds'], timed_kernel_ms=kms,
                       gpu_busy_fraction_timed_kernels=kms * 1e-3 / r['seconds'], frac_mfma_f32=r['roofline']['frac'],
                       ms_per_segment_sweep=1e3 * r['seconds'] / max(1, sweeps))
            row['kernel_ms_total'] = {k: round(v['ms_per_launch'] * v['launches'], 2) for k, v in r['kernels'].items() if 'launches' in v}
            row['kernel_launches'] = {k: v['launches'] for k, v in r['kernels'].items() if 'launches' in v}
            rows.append(row)
            print(json.dumps(row), flush=True)
    by = {}
    for r in rows:
        by.setdefault(r['workload'], []).append(r)
    for name, rs in by.items():
        if len(rs) == 2:
            print("%s: %d segments run at %.2f of the one-segment rate (%.3g vs %.3g instance-sweeps/s); GPU busy %.2f vs %.2f"
                  % (name, rs[0]['segments'], rs[0]['instance_sweeps_per_s'] / rs[1]['instance_sweeps_per_s'], rs[0]['instance_sweeps_per_s'],
                     rs[1]['instance_sweeps_per_s'], rs[0]['gpu_busy_fraction_timed_kernels'], rs[1]['gpu_busy_fraction_timed_kernels']))


if __name__ == '__main__':
    main()
