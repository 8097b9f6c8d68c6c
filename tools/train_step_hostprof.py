"""cProfile of the HOST side of a training step (GPU box): which Python functions the 51 ms of enqueue time go to.   python tools/train_step_hostprof.py"""
import cProfile
import importlib
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import train_step_probe as tp  # noqa: E402


def main():
    once, dev = tp.build_step()
    for _ in range(3):
        once()
    torch.cuda.synchronize(dev)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        once()
    torch.cuda.synchronize(dev)
    pr.disable()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
