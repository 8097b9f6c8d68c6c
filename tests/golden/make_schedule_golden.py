"""gamma tables of the predefined noise schedules as the REFERENCE builds them (PredefinedNoiseSchedule, variational_diffusion.py:205-250):
cosine, polynomial_2 (production) and polynomial_3, T = 1000 and 250.   python tests/golden/make_schedule_golden.py -> gamma_tables.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402

_, vd, _ = rh.import_reference()
out = {}
for name in ("cosine", "polynomial_2", "polynomial_3"):
    for T in (1000, 250):
        out[f"{name}__{T}"] = vd.PredefinedNoiseSchedule(name, T, 1e-5, verbose=False).gamma.detach().numpy()
np.savez_compressed(os.path.join(HERE, "gamma_tables.npz"), **out)
print({k: v.shape for k, v in out.items()})
