import cProfile, pstats, sys, os, io
sys.argv = ["training_script_bench.py", "--epochs", "3"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
# run the bench once to warm everything (3 epochs), then profile two more epochs through its globals
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "training_script_bench.py"), run_name="__main__")
import torch
vt, m, songs, s = g["vae_training"], g["m"], g["songs"], g["s"]
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for ep in (3, 4):
    vt.run_epoch(m, songs, s, ep, train=True)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
