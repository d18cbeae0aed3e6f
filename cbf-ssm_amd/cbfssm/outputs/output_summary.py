"""`OutputSummary(out_dir)`: collects the final RMSE of several runs (`add_outputs(outputs)`) and writes
`summary.txt` with the per-run values, their mean and standard deviation; the driving script is copied next to it as
`main.py` (interface of the reference's cbfssm/outputs/output_summary.py:7-31)."""
import os
import shutil
import sys
import numpy as np


class OutputSummary:

    def __init__(self, out_dir):
        self.out_dir = out_dir
        self.rmse_all = []
        os.makedirs(out_dir, exist_ok=True)
        script = sys.argv[0] if sys.argv else ''
        self._writer = int(os.environ.get('RANK', '0')) == 0      # data parallel: rank 0 writes the files
        if script and os.path.isfile(script) and self._writer:
            shutil.copyfile(os.path.abspath(script), os.path.join(out_dir, 'main.py'))

    def add_outputs(self, outputs):
        self.rmse_all.append(outputs.get_last_rmse())

    def write_summary(self):
        runs = [r for r in self.rmse_all]
        if not runs or runs[0] is None:
            print("RMSE summary skipped")
            return
        values = np.asarray(runs, dtype=np.float64)
        lines = ['RMSE', '====', '', 'Runs:']
        lines += ['  %f' % v for v in values]
        lines += ['Mean: %f' % values.mean(), 'Std:  %f' % values.std()]
        if not self._writer:
            return
        with open(os.path.join(self.out_dir, 'summary.txt'), 'w') as fh:
            fh.write('\n'.join(lines) + '\n')
