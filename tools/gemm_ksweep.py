"""Fixed cost vs per-K-step cost of the fp32 GEMM: one round of 256 tiles (M = N = 2048) and five
rounds (M = 10240), K swept."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as g
for M in (2048, 10240):
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        g.run("M=%d K=%d" % (M, K), M, 2048, K, 0, 0, 0, iters=20)
