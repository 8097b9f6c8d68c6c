"""Golden RePaint schedules from the REFERENCE's EquivariantVariationalDiffusion.get_repaint_schedule
(src/models/components/variational_diffusion.py:1548-1578), a pure function of three integers.

    python tests/golden/make_repaint_golden.py       (build container only)   ->  tests/golden/repaint_schedule.json
"""
import itertools
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402


def main():
    _, vd, _ = rh.import_reference()
    fn = vd.EquivariantVariationalDiffusion.get_repaint_schedule
    fn = getattr(fn, "__wrapped__", fn)
    cases = []
    for r, j, t in itertools.product((1, 2, 3, 10), (1, 2, 3, 7, 10, 40), (1, 2, 5, 6, 10, 21, 100, 1000)):
        cases.append(dict(resamplings=r, jump_length=j, num_timesteps=t, schedule=fn(None, r, j, t)))
    with open(os.path.join(HERE, "repaint_schedule.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print(len(cases), cases[7], cases[-1]["schedule"][:5], sum(cases[-1]["schedule"]))


if __name__ == "__main__":
    main()
