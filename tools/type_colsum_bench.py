"""Micro-benchmark of dvae_type_colsum at the cfg 2 shape (8192 rows, Hp = 512, 8 types; HIP events)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from _dvae_pkg import dvae_amd
L = dvae_amd._lib; lib = L.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n, Hp, T = 8192, 512, 8
g = torch.Generator().manual_seed(0)
types = torch.randint(0, T, (n,), generator=g).int().cuda()
dgh = torch.randn(n, 3 * Hp, generator=g).cuda(); dgin = torch.randn(n, Hp, generator=g).cuda()
off = torch.zeros(T + 1, dtype=torch.int32).cuda(); idx = torch.zeros(n, dtype=torch.int32).cuda()
gi = torch.zeros(32, 3 * Hp).cuda(); bhn = torch.zeros(Hp).cuda()
L.check(lib.dvae_type_csr(types.data_ptr(), n, None, 0, T, off.data_ptr(), idx.data_ptr(), s))
def f(): L.check(lib.dvae_type_colsum(dgh.data_ptr(), dgin.data_ptr(), off.data_ptr(), idx.data_ptr(), T, gi.data_ptr(), bhn.data_ptr(), n, Hp, s))
for _ in range(20): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200): f()
e1.record(); torch.cuda.synchronize()
print("dvae_type_colsum %d rows x Hp %d, %d types: %.2f us" % (n, Hp, T, e0.elapsed_time(e1) / 200 * 1e3))
