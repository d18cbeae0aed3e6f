"""Per-kernel MFMA utilisation from a rocprofv3 --pmc pass (profiles/tools/collect_mfma_util.sh).

    mfma_util_summary.py <workload> <pmc dir>

For every time-loop kernel (largest dispatches of each name = the full launches): mean counter values per dispatch and
  issue share   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs)     matrix-unit busy share of the CU-busy time
  MOPS check    = SQ_INSTS_VALU_MFMA_MOPS_F64 against the instruction count the kernel's step structure predicts
(gfx950: one v_mfma_f64_16x16x4_f64 = 64 issue cycles on a SIMD, csrc/probe/mfma_f64_probe.hip; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles, MI355X_MICROARCH.md 'cycle constants'.)"""
import collections
import csv
import glob
import sys

wl, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        if 'cbfssm' not in name:
            continue
        key = (name.split('(')[0][:90], row.get('Grid_Size', ''), row.get('Workgroup_Size', ''))
        acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
print('workload', wl)
for (name, grid, wg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', [0]))):
    if not any(t in name for t in ('pass_kernel', 'rev_kernel', 'stash_contract')):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    print('%s  grid=%s wg=%s  dispatches=%d' % (name, grid, wg, n))
    for c in sorted(m):
        print('    %-32s %.6g' % (c, m[c]))
    busy_cu = m.get('SQ_BUSY_CU_CYCLES')
    mfma = m.get('SQ_VALU_MFMA_BUSY_CYCLES')
    if busy_cu and mfma:
        print('    -> matrix-unit busy share of CU-busy cycles (4 SIMDs per CU): %.3f' % (mfma / (4.0 * busy_cu)))
    if m.get('SQ_INSTS_VALU_MFMA_MOPS_F64') and mfma:
        print('    -> MFMA busy cycles per F64 MOPS unit: %.3f' % (mfma / m['SQ_INSTS_VALU_MFMA_MOPS_F64']))
