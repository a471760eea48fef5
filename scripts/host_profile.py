"""cProfile of a bench leg's host side: python scripts/host_profile.py <bench.py arguments...> -- top functions by own time."""
import cProfile, pstats, sys, io, runpy
sys.argv = ["bench.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:48]), file=sys.stderr)
