import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nnet_train_bench as nb
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nb.measure("cfg2", b, 16, 30, 8)
pr = cProfile.Profile(); pr.enable()
r = nb.measure("cfg2", b, 16, 60, 8)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
