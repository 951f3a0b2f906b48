"""cProfile of an example's host side (where does a training iteration's wall time go?).
    python tools/profile_host_step.py laplace2d epochs=2000 eval_freq=2000 log_freq=1000
    python tools/profile_host_step.py allen_cahn_plain epochs=1 iters_per_epoch=600 log_freq=300"""
import cProfile
import importlib
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1]
sys.argv = [name] + sys.argv[2:] + ["output_dir=/tmp/ppsci_host_profile"]
ex = importlib.import_module(f"examples.{name}")
pr = cProfile.Profile()
pr.enable()
ex.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:8000])
