#!/usr/bin/env python3
"""Condenses tools/profile_secondary.sh's rocprofv3 directories: per workload the kernel stats (top kernels) and, per kernel, the PMC
counters per dispatch plus the derived ratios DESIGN.md quotes (MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES; the issue share of
the other VALU instructions at 2 cycles each -- an fp32 MFMA occupies the SIMD's VALU issue, profiles/r06_mfma_valu_overlap.txt, so the two
ADD; wait share).  Writes <root>/neural_pmc.json = the per-launch instruction counts of the hidden-128 kernels, which bench.py's `issue_bound`
reads (copied to profiles/<tag>_neural_pmc.json)."""
import json
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
neural_pmc = {}
for work in sorted(os.listdir(root)):
    wd = os.path.join(root, work)
    if not os.path.isdir(wd):
        continue
    print("==== %s ====" % work)
    avg_us = {}
    for f in sorted(glob.glob(os.path.join(wd, 'trace', '**', '*kernel_stats.csv'), recursive=True)):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        for r in rows:
            avg_us.setdefault(r.get('Name', '')[:64], float(r.get('AverageNs', 0)) / 1e3)
        for r in rows[:10]:
            print("  %-64s calls=%-5s avg_us=%-10.1f pct=%s" % (r.get('Name', '')[:64], r.get('Calls'), float(r.get('AverageNs', 0)) / 1e3, r.get('Percentage')))
    per = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(set))
    for f in sorted(glob.glob(os.path.join(wd, 'pmc*', '**', '*counter_collection.csv'), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get('Kernel_Name', '')[:64]
                c = r.get('Counter_Name')
                per[k][c] += float(r.get('Counter_Value', 0) or 0); cnt[k][c].add(r.get('Dispatch_Id'))
    for k in sorted(per, key=lambda x: -per[x].get('SQ_BUSY_CU_CYCLES', per[x].get('SQ_WAVE_CYCLES', 0.0)))[:6]:
        v = {c: per[k][c] / max(1, len(cnt[k][c])) for c in per[k]}
        print("  [pmc] %s" % k)
        print("        " + "  ".join("%s=%.4g" % (c, v[c]) for c in sorted(v)))
        if v.get('SQ_BUSY_CU_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the four matrix pipes of a CU (= 64 x SQ_INSTS_MFMA for v_mfma_f32_32x32x2_f32),
            # SQ_BUSY_CU_CYCLES counts cycles per busy CU: the pipe's busy fraction is their ratio / 4 SIMDs
            print("        MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = %.3f   (cycles per MFMA instruction: %.1f)"
                  % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_BUSY_CU_CYCLES']), v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, v.get('SQ_INSTS_MFMA', 0.0))))
            if 'SQ_INSTS_VALU' in v:
                other = v['SQ_INSTS_VALU'] - v.get('SQ_INSTS_MFMA', 0.0)          # SQ_INSTS_VALU counts the MFMAs too
                print("        other VALU issue at 2 cycles per wave-instruction = 2 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) / (4 x SQ_BUSY_CU_CYCLES) = %.3f   (MFMA + VALU = %.3f of the SIMD cycles)"
                      % (2.0 * other / (4.0 * v['SQ_BUSY_CU_CYCLES']), (v['SQ_VALU_MFMA_BUSY_CYCLES'] + 2.0 * other) / (4.0 * v['SQ_BUSY_CU_CYCLES'])))
            if work == 'neural128':
                for key, prefix, unit in (('gru', 'void k_gru_pipe<65', 'edges'), ('agg_pre', 'void k_agg_pre_wave<65', 'edges'), ('agg_post', 'void k_agg_post_pf<26', 'edges'),
                                          ('predict_head', 'void k_predict_rows_pf<', 'variables')):
                    if k.startswith(prefix) and 'SQ_INSTS_MFMA' in v:
                        # shader clock under this kernel's load: busy cycles of a CU / the kernel's duration (all 256 CUs busy for the whole launch: a lower
                        # bound of the clock).  The fp32-MFMA kernels run throttled: 2.1-2.3 GHz, not the 2.4 GHz the peak is quoted at.
                        us = avg_us.get(k)
                        neural_pmc[key] = {'kernel': k[5:].split('(')[0], 'unit': unit, 'SQ_INSTS_VALU': v['SQ_INSTS_VALU'], 'SQ_INSTS_MFMA': v['SQ_INSTS_MFMA'],
                                           'SQ_VALU_MFMA_BUSY_CYCLES': v['SQ_VALU_MFMA_BUSY_CYCLES'], 'SQ_BUSY_CU_CYCLES': v['SQ_BUSY_CU_CYCLES'],
                                           'clock_ghz_profiled': (v['SQ_BUSY_CU_CYCLES'] / 256.0 / (us * 1e-6) / 1e9) if us else None,
                                           'SQ_WAIT_ANY_over_SQ_WAVE_CYCLES': (v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']) if v.get('SQ_WAVE_CYCLES') else None}
        if v.get('SQ_WAVE_CYCLES') and 'SQ_WAIT_ANY' in v:
            print("        SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f" % (v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']))

if neural_pmc:
    json.dump({'source': 'tools/profile_secondary.sh: rocprofv3 --pmc passes of bench.py --workload neural --hidden 128 (np-nd-np on 5 000 x n=200); counters per launch',
               'edges': 12600000, 'variables': 999999, 'kernels': neural_pmc}, open(os.path.join(root, 'neural_pmc.json'), 'w'), indent=1)
