"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as g
M, N, K, akm, bkm, acc = [int(x) for x in sys.argv[1:7]]
g.run("one", M, N, K, akm, bkm, acc, iters=5)
