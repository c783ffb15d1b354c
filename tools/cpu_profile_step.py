"""Host-side profile of the training step (cProfile over the timed steps of bench.py's own step function).
usage: cpu_profile_step.py [batch] [steps]"""
import cProfile, pstats, sys, io
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]] + sys.argv[1:]
batch = sys.argv[1] if len(sys.argv) > 1 else "12"
steps = sys.argv[2] if len(sys.argv) > 2 else "30"
import bench
sys.argv = ["bench.py", "--batch", batch, "--steps", steps, "--warmup", "3", "--no-cpu-baseline", "--no-batch12", "--no-fp32-batch12", "--no-stage2"]
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
