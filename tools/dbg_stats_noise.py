"""How reproducible are the densification statistics of a 10-iteration single-process trajectory? (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_scaffold_dp_gpu import _setup, _densify_params
dev = torch.device("cuda:0")
runs = []
for rep in range(2):
    kfs, model, step, gts = _setup(dev, False, _densify_params())
    dens = step.densifier
    for it in range(1, 11):
        step.iteration += 1
        for k in range(2):
            step._forward_backward(kfs[k][1], gts[k])
            if dens.p.start_stat < step.iteration:
                dens.training_statis(step.neural, step.visible_radii, step.engine.radii, step.engine.dL_dmean2D)
        if it == 10:
            break
        step.world = 2
        step._adam(model.adam_groups(step.learning_rates(step.iteration)), step._mlp_count, None)
        step.world = 1
    torch.cuda.synchronize()
    runs.append(({n: dens.stat(n).cpu().numpy().copy() for n in dens.STAT_NAMES}, model.params.cpu().numpy().copy()))
for n in runs[0][0]:
    a, b = runs[0][0][n], runs[1][0][n]
    bad = np.abs(a - b) > 1e-3 * np.abs(b) + 1e-4 * np.abs(b).max()
    print(n, "bad frac", bad.mean(), "max", np.abs(b).max(), "sum a/b", a.sum(), b.sum(), "corr", np.corrcoef(a.ravel(), b.ravel())[0, 1])
pa, pb = runs[0][1], runs[1][1]
print("params differ frac", (pa != pb).mean(), "max diff", np.abs(pa - pb).max())
