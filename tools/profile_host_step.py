import cProfile, pstats, sys, os, io
sys.path.insert(0, "/root/repo")
sys.argv = ["x", "epochs=1", "iters_per_epoch=600", "log_freq=300", "output_dir=/tmp/o9"]
import examples.allen_cahn_plain as ex
pr = cProfile.Profile()
pr.enable()
ex.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
