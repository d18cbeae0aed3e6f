"""Multi-run RMSE summary (cbfssm/outputs/output_summary.py:7-31): copies the driving script, writes summary.txt."""
import os
import sys
import numpy as np
from shutil import copyfile


class OutputSummary:

    def __init__(self, out_dir):
        self.out_dir = out_dir
        self.rmse_all = []
        os.makedirs(self.out_dir, exist_ok=True)
        src = os.path.abspath(sys.argv[0]) if sys.argv and sys.argv[0] else None
        if src and os.path.isfile(src):
            copyfile(src, self.out_dir + '/main.py')

    def add_outputs(self, outputs):
        self.rmse_all.append(outputs.get_last_rmse())

    def write_summary(self):
        if self.rmse_all and self.rmse_all[0] is not None:
            rmse_all = np.asarray(self.rmse_all, dtype=np.float64)
            with open(self.out_dir + '/summary.txt', 'w') as f:
                f.write("RMSE\n====\n\nRuns:\n")
                for val in rmse_all:
                    f.write("  %f\n" % val)
                f.write("Mean: %f\n" % np.mean(rmse_all))
                f.write("Std:  %f\n" % np.std(rmse_all))
        else:
            print("RMSE summary skipped")
