"""ISA audit of one kernel: where its vector-memory waits sit and what they really wait for.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps=obj -c cbf-ssm_amd/csrc/rev_nb7.hip -o /tmp/isa/rev_nb7.o
    python3 profiles/tools/isa_wait_audit.py /tmp/isa/rev_nb7-hip-amdgcn-amd-amdhsa-gfx950.s <mangled kernel name> [trace|waits]

`trace`: loads / stores / scratch accesses / s_waitcnt vmcnt / s_barrier in program order with the MFMAs counted per run --
which MFMAs sit between which barriers (how the extra wave's MFMAs were found below two s_barriers), where a load is followed
at once by its wait.
`waits` (default): for every `s_waitcnt vmcnt(N)` the operations it waits for.  vmcnt retires IN ORDER and counts stores and
scratch accesses, so a wait for an old load also waits for everything issued before the N youngest operations: L = global
load, W = global store, S = scratch access (a spill reload).  Lines flagged `<==` wait for something issued a few
instructions earlier (a full memory latency); a wait whose list holds W's behind the load it is for waits for store
completion.  Static program order, not loop aware: read it per barrier interval.  """
import re
import sys


def kernel_lines(path, name):
    s = open(path).read()
    i = s.index(name + ':')
    j = s.index('.end_amdhsa_kernel', i)
    return s[i:j].split('\n')


def trace(lines):
    out, run = [], 0
    for n, l in enumerate(lines):
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            run += 1
            continue
        if op.startswith('s_barrier') or 'vmcnt' in t or op.startswith('global_') or op.startswith('scratch_'):
            if run:
                out.append('       ... %d MFMA' % run)
                run = 0
            out.append('%6d %s' % (n, t[:100]))
    if run:
        out.append('       ... %d MFMA' % run)
    return out


def waits(lines):
    pend, out = [], []
    for n, l in enumerate(lines):
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('global_load'):
            pend.append((n, 'L'))
        elif op.startswith('scratch_'):
            pend.append((n, 'S'))
        elif op.startswith('global_store') or op.startswith('global_atomic'):
            pend.append((n, 'W'))
        elif op == 's_barrier':
            out.append('%6d ---- s_barrier' % n)
        m = re.search(r'vmcnt\((\d+)\)', t)
        if m and op == 's_waitcnt':
            keep = int(m.group(1))
            waited = pend[:len(pend) - keep] if keep < len(pend) else []
            if waited:
                flag = '   <== waits for an operation issued <= 6 lines above' if any(n - ln <= 6 for ln, _ in waited) else ''
                out.append('%6d vmcnt(%d): waits for %d [%s], oldest %d / youngest %d lines back%s' %
                           (n, keep, len(waited), ''.join(k for _, k in waited), n - waited[0][0], n - waited[-1][0], flag))
            pend = pend[len(pend) - keep:] if keep < len(pend) else pend
    return out


if __name__ == '__main__':
    ls = kernel_lines(sys.argv[1], sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else 'waits'
    m = re.search(r'\.amdhsa_next_free_vgpr\s+(\d+)', '\n'.join(ls))
    sc = re.search(r'\.amdhsa_private_segment_fixed_size\s+(\d+)', '\n'.join(ls))
    print('# VGPRs %s, scratch %s B/lane' % (m.group(1) if m else '?', sc.group(1) if sc else '?'))
    print('\n'.join(trace(ls) if mode == 'trace' else waits(ls)))
