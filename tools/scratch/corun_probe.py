"""Scratch: the resident tridiagonalisation of 32 order-512 matrices with and without an fp64-MFMA burner of <= 64
registers per lane on another stream (corun_probe.hip).  usage: python tools/scratch/corun_probe.py [B] [lds_kb] [wgs]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imgcompressionmps_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lds_kb = int(sys.argv[2]) if len(sys.argv) > 2 else 96
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 256
n, k = 512, 64
lib = _lib.load()
burn = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libcorun.so"))
burn.corun_burn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
burn.corun_burn32.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
FP32 = os.environ.get("CORUN_FP32") is not None  # the fp32-MFMA burner instead of the fp64 one
dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(1)
a = torch.randn((B, n + 64, n), dtype=torch.float64, device=dev, generator=gen)
a = a * torch.logspace(0, -4, n, dtype=torch.float64, device=dev)[None, None, :]
g0 = torch.bmm(a.transpose(1, 2), a).contiguous()
sizes = _lib.i64_array([n] * B)
v = torch.zeros_like(g0)
w = torch.zeros((B, n), dtype=torch.float64, device=dev)
nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
src = torch.rand(4096, dtype=torch.float32, device=dev)
out = torch.zeros(wgs * 256, dtype=torch.float64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def values(stream):
    _lib.check(lib.ndmps_syevd_topk_values_f64(B, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k,
                                               ws.data_ptr(), nb, stream.cuda_stream))


def burner(iters):
    fn = burn.corun_burn32 if FP32 else burn.corun_burn
    assert fn(wgs, iters, lds_kb * 1024, src.data_ptr(), out.data_ptr(), s2.cuda_stream) == 0


def ev():
    return torch.cuda.Event(enable_timing=True)


values(s1)
burner(100)
torch.cuda.synchronize()
# alone
t = []
for _ in range(3):
    e0, e1 = ev(), ev()
    e0.record(s1)
    values(s1)
    e1.record(s1)
    torch.cuda.synchronize()
    t.append(e0.elapsed_time(e1))
t_alone = min(t)
iters = 8000
e0, e1 = ev(), ev()
e0.record(s2)
burner(iters)
e1.record(s2)
torch.cuda.synchronize()
b_alone = e0.elapsed_time(e1)
flop = wgs * 4 * iters * 32 * (4096 if FP32 else 2048)
print(f"alone: tridiagonalisation + eigenvalues of {B} matrices {t_alone:.3f} ms; burner ({wgs} workgroups, {lds_kb} KB LDS) "
      f"{b_alone:.3f} ms = {flop / b_alone / 1e9:.1f} TFLOP/s", flush=True)
# together: the burner first, then three reductions back to back while it runs
for rep in range(2):
    b0, b1 = ev(), ev()
    es = [ev() for _ in range(4)]
    b0.record(s2)
    burner(iters)
    b1.record(s2)
    es[0].record(s1)
    for i in range(3):
        values(s1)
        es[i + 1].record(s1)
    torch.cuda.synchronize()
    print(f"together: burner {b0.elapsed_time(b1):.3f} ms ({flop / b0.elapsed_time(b1) / 1e9:.1f} TFLOP/s), reductions "
          + " ".join(f"{es[i].elapsed_time(es[i + 1]):.3f}" for i in range(3)) + " ms; resident launches given up: "
          f"{lib.ndmps_syevd_topk_team_fallbacks()}", flush=True)
ref = torch.linalg.eigvalsh(g0).flip(-1)
print(f"|dw|/w0 {float((w[:, :k] - ref[:, :k]).abs().max() / ref.abs().max()):.1e}")
