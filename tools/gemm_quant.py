"""Wave-quantization probe: time the forward GEMM (N=2048, K=512) for row counts around
multiples of the chip's tile slots."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as g
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for M in (2048, 4096, 4224, 6144, 8192, 8320, 9251, 10240, 12288, 16384, 32768):
    g.run("M=%d tiles128=%d" % (M, -(-M // 128) * -(-N // 128)), M, N, K, 0, 0, 0, iters=10)
