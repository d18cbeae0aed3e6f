"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV (<prefix>_kernel_trace.csv): every launch between two
markers (by default the prepare kernels of consecutive steps) with its start offset, duration and queue -- what runs
next to what on the two streams of a train step.  Usage:
    python3 profiles/tools/step_timeline.py <kernel_trace.csv> [step_index=3] [marker=prepare_kernel]"""
import csv
import sys


def short(name):
    n = name.replace('void ', '').replace('cbfssm::', '')
    for a, b in (('at::native::', ''), ('vectorized_elementwise_kernel', 'elementwise')):
        n = n.replace(a, b)
    return n[:64]


def main():
    path = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    marker = sys.argv[3] if len(sys.argv) > 3 else 'prepare_kernel'
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    if len(marks) < step + 2:
        print('only %d markers' % len(marks))
        return
    i0, i1 = marks[step], marks[step + 1]
    t0 = int(rows[i0]['Start_Timestamp'])
    print('step %d: %d launches, %.3f ms from its first kernel to the next step\'s first' %
          (step, i1 - i0, (int(rows[i1]['Start_Timestamp']) - t0) * 1e-6))
    qs = {}
    busy = {}
    for r in rows[i0:i1]:
        q = r.get('Queue_Id', '?')
        qs.setdefault(q, len(qs))
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        busy[q] = busy.get(q, 0) + (e - s)
        print('%9.3f %9.3f  q%d  %-64s grid %s wg %s' % (s * 1e-6, (e - s) * 1e-6, qs[q], short(r['Kernel_Name']),
                                                        r.get('Grid_Size', r.get('Grid_Size_X', '?')),
                                                        r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))))
    for q, b in busy.items():
        print('queue q%d busy %.3f ms' % (qs[q], b * 1e-6))


if __name__ == '__main__':
    main()
