"""HBM bytes per launch of the four time-loop kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB),
written as profiles/<round>/traffic.json (bench.py reads it into roofline.traffic).
usage: make_traffic_json.py <workload> <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, re, sys, collections

wl, fdir, wdir, out = sys.argv[1:5]


STASH_KINDS = set()      # logical kernels that run as several launches per step (stash mode): summed per step, not max'd


def kind(name):
    if 'stash_contract' in name and 'reduce' not in name:
        STASH_KINDS.add('stash_contraction')
        return 'stash_contraction'
    m = re.search(r'(rev_kernel|pass_kernel_skew|pass_kernel)<([^>]*)>', name)
    if not m:
        return None
    args = [x.strip() for x in m.group(2).split(',')]
    fam = m.group(1)
    mode = int(args[5] if fam == 'rev_kernel' else args[4])
    base = 'backward_pass' if mode == 1 else 'forward_pass'
    if fam == 'rev_kernel' and args[4] == 'true':       # stash mode: time-chunked launches
        STASH_KINDS.add(base + '_adjoint')
    return base + ('_adjoint' if fam == 'rev_kernel' else '')


def collect(path, counter):
    acc = collections.defaultdict(list)
    nsteps = 0
    for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            if 'prepare_kernel' in row['Kernel_Name']:
                nsteps += 1                               # one per ELBO evaluation of the run
            k = kind(row['Kernel_Name'])
            if k:
                acc[k].append(float(row['Counter_Value']))
    acc['_nsteps'] = nsteps
    return acc


fe, wr = collect(fdir, 'FETCH_SIZE'), collect(wdir, 'WRITE_SIZE')
res = {}
nsteps = max(1, fe.pop('_nsteps'))
wr.pop('_nsteps', None)
for k in sorted(fe):
    # the largest dispatches are the full launches (bench.py's per-kernel timing); the smaller ones are the pieces of the
    # chain-group split inside a whole step.  Stash-mode kernels run as several launches per step: all dispatches of the
    # run summed, divided by the number of train-step evaluations in it (one prepare launch each)
    if k in STASH_KINDS:
        f, w = sum(fe[k]) / nsteps, sum(wr.get(k, [0.0])) / nsteps
    else:
        f, w = max(fe[k]), max(wr.get(k, [0.0]))
    res[k] = int((2.0 * f + w) * 1024)
    print('%-24s fetch 2 x %.1f MB  write %.1f MB  (%d dispatches)' % (k, f * 1024 / 1e6, w * 1024 / 1e6, len(fe[k])))
res['_note'] = ('bytes per full launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 from separate rocprofv3 --pmc passes of '
                '`bench.py --workload %s --steps 1 --warmup 1` (profiles/tools/collect_traffic.sh).  FETCH_SIZE x 2 as '
                'MI355X_MICROARCH.md (HBM) prescribes for coalesced streams: these kernels read 128..512 contiguous bytes '
                'per wave-instruction (128-B requests tallied at 64 B).  Calibrated on the eval-mode backward pass, whose '
                'compulsory reads are known (noise 41 MB + u, y 7 MB = 48 MB; FETCH_SIZE reports 22 MB), and on '
                'loglik_moments_kernel (147 MB compulsory reads): 156 MB reported while it read 8 B per lane at a 112-B '
                'stride (no correction applies), 83 MB reported now that a wave reads contiguous 112-B runs (2 x 83 = 166 '
                'MB: partial 128-B lines make the factor 1.8 rather than 2 there).' % wl.replace(':', ' --mode '))
try:
    allj = json.load(open(out))
except Exception:
    allj = {}
allj[wl] = res
json.dump(allj, open(out, 'w'), indent=1)
