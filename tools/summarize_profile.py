#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel stats + PMC csv files) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else ''      # 'fast_': the passes of the opt-in fast build (profile_bench.sh)


def find(pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find('*kernel_stats.csv'):
    if '/%strace/' % prefix not in f:
        continue
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get('Name', '')[:70], r.get('Calls'), r.get('TotalDurationNs'),
                                                                r.get('AverageNs'), r.get('Percentage')))
print()
print("== PMC counters per kernel (sum over dispatches / dispatches) ==")
for d in ('pmc1', 'pmc2', 'pmc5', 'pmc6', 'pmc7', 'pmc3', 'pmc4'):
    for f in find('*counter_collection.csv'):
        if '/%s%s/' % (prefix, d) not in f:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get('Kernel_Name', '')[:60]
                acc[k][r.get('Counter_Name')] += float(r.get('Counter_Value', 0) or 0)
                cnt[k].add(r.get('Dispatch_Id'))
        for k in sorted(acc, key=lambda x: -sum(acc[x].values()))[:4]:
            n = max(1, len(cnt[k]))
            print("[%s] %-60s dispatches=%d" % (d, k, n))
            for c, v in sorted(acc[k].items()):
                print("      %-28s %.4g per dispatch" % (c, v / n))

# HBM traffic of the solver kernel per launch, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE under-reports
# wide reads by 2x on gfx950; both counters are in KiB)
import json
vals = {}
for d, name in (('pmc3', 'FETCH_SIZE'), ('pmc4', 'WRITE_SIZE')):
    for f in find('*counter_collection.csv'):
        if '/%s%s/' % (prefix, d) not in f:
            continue
        tot, disp = 0.0, set()
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if 'k_sp_solve_lds<false, false' in r.get('Kernel_Name', '') and r.get('Counter_Name') == name:
                    tot += float(r.get('Counter_Value', 0) or 0); disp.add(r.get('Dispatch_Id'))
        if disp:
            vals[name] = tot / len(disp)
if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
    out = dict(FETCH_SIZE_KiB_per_launch=vals['FETCH_SIZE'], WRITE_SIZE_KiB_per_launch=vals['WRITE_SIZE'],
               k_sp_solve_lds_bytes_per_launch=(2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0,
               note='2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; average over the launches of k_sp_solve_lds<false, false, false> (one per chunk of PDP_SOLVE_CHUNK iterations, `launches_per_call` of them per 100-sweep call; the poison replay is k_sp_solve_lds<false, true, false>)')
    # wave-level VALU instructions per launch of the same kernel (pmc1 of this build and of the fast build): bench.py prices them against the
    # SIMDs' issue slots (roofline.valu_issue)
    for pre, key in (('', 'SQ_INSTS_VALU_per_launch'), ('fast_', 'SQ_INSTS_VALU_per_launch_fast_build')):
        for f in find('*counter_collection.csv'):
            if '/%spmc1/' % pre not in f:
                continue
            tot, disp, calls = 0.0, set(), set()
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if 'k_sp_solve_lds<false, false' in r.get('Kernel_Name', '') and r.get('Counter_Name') == 'SQ_INSTS_VALU':
                        tot += float(r.get('Counter_Value', 0) or 0); disp.add(r.get('Dispatch_Id'))
                    if 'k_solve_import' in r.get('Kernel_Name', ''):
                        calls.add(r.get('Dispatch_Id'))          # one import per pdp_sp_solve call
            if disp:
                out[key] = tot / len(disp)
                if not pre:
                    out['launches_per_call'] = len(disp) / float(max(1, len(calls)))
    out['source'] = 'bash tools/profile_bench.sh <tag>: rocprofv3 --kernel-trace --pmc passes of python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-build; FETCH_SIZE and WRITE_SIZE in passes of their own'
    json.dump(out, open(os.path.join(root, prefix + 'pmc_traffic.json'), 'w'), indent=1)
    print()
    print('== HBM traffic of k_sp_solve_lds per launch:', json.dumps(out))
