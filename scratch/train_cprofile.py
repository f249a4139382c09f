import cProfile, pstats, io, os, sys, runpy
sys.argv = ["train_prof.py"]
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_prof.py")).read()
g = {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_prof.py")}
code = compile(src.split("# steady state")[0], "train_prof.py", "exec")
exec(code, g)
pr.enable()
for _ in range(5): g["step"]()
g["torch"].cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
