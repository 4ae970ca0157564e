"""worst gradient disagreement (time-parallel vs sequential kernels) per state width and decade of the conditioning score,
from the rows tools/gp_cond_bins.py saved: python tools/gp_cond_table.py gpurun_out/gp_cond_bins_*.npy"""
import sys

import numpy as np

rows = np.concatenate([np.load(f) for f in sys.argv[1:]])
kappa = (1.0 + rows[:, 0]) * rows[:, 1]
J = rows[:, 4].astype(int)
print("draws", len(rows))
dec = list(range(0, 11))
print("%8s" % "J" + "".join("%14s" % f"1e{d}" for d in dec[:-1]))
for name, sel in (("1-2", J <= 2), ("3", J == 3), ("4", J == 4), ("5-6", J >= 5)):
    line = "%8s" % name
    for d in dec[:-1]:
        m = sel & (kappa >= 10.0 ** d) & (kappa < 10.0 ** (d + 1))
        line += "%14s" % (f"{rows[m, 3].max():.0e}/{np.median(rows[m, 3]):.0e} ({m.sum()})" if m.any() else "-")
    print(line)
