import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as g
for (M,N,K) in [(8192,2048,128),(8192,2048,256),(8192,2048,512),(8192,2048,1024),(8192,2048,2048),(4096,2048,512),(2048,2048,512),(16384,2048,512)]:
    g.run("big  ", M, N, K, 0, 0, 0, iters=20)
