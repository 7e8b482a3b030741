"""Host-side profile of the LogTrainer loop (phase 1, synthetic data): where does the wall time per step go?"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan import cli
argv = ["--dataset", "cifar10", "--num_data", "4096", "--max_steps", "60", "--loss_type", "ns", "--work_dir",
        os.path.join(ROOT, "gpurun_out", "exp_prof"), "--exp_name", "p", "--no_save_logits", "--save_steps", "1000"]
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
cli.phase1(argv)
pr.disable()
torch.cuda.synchronize()
print(f"total {time.perf_counter() - t0:.2f} s for 60 steps")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
print(s.getvalue()[:7000])
