"""Fixed cost vs per-K-step cost of the fp32 GEMM, K swept:
  64x64 tiles, one per CU (1024 x 1024), 1.5 per CU (1024 x 1536), and 128x128 tiles one round
  (2048 x 2048) / five rounds (10240 x 2048)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as g
for M, N in ((1024, 1024), (1024, 1536), (2048, 2048), (10240, 2048)):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        g.run("M=%d N=%d K=%d" % (M, N, K), M, N, K, 0, 0, 0, iters=20)
