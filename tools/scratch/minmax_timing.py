"""Scratch: where does the host spend the 3.4 ms between the cores' fp32 copies and the min/max launch of a config-5
step?  Times the statements of utils/filetools.minmax_many on the calls with five small tensors."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from imgcompressionmps_amd import _lib  # noqa: E402
from imgcompressionmps_amd.utils import filetools as ft  # noqa: E402

acc = {}
orig = ft.minmax_many


def timed(tensors, with_sumsq=False):
    tensors = list(tensors)
    if len(tensors) < 2:
        return orig(tensors, with_sumsq)
    lib = _lib.load()
    t = [time.perf_counter()]
    ts = [x.to(torch.float32) if x.dtype is not torch.float32 else x for x in tensors]
    t.append(time.perf_counter())
    count = len(ts)
    nbytes = lib.ndmps_minmax_many_workspace_bytes(count)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ts[0].device)
    t.append(time.perf_counter())
    ptrs = (C.c_void_p * count)(*[x.data_ptr() for x in ts])
    lens = _lib.i64_array([x.numel() for x in ts])
    out = (C.c_float * (2 * count))()
    ss = (C.c_double * count)()
    t.append(time.perf_counter())
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    _lib.check(lib.ndmps_minmax_many_f32(count, ptrs, lens, out, ss, ws.data_ptr(), nbytes, _lib.stream_ptr()))
    t.append(time.perf_counter())
    for k, name in enumerate(["to_fp32", "workspace", "ctypes arrays", "synchronize (what was still running)", "library call"]):
        acc.setdefault(name, []).append((t[k + 1] - t[k]) * 1e3)
    mm = [(float(out[2 * i]), float(out[2 * i + 1])) for i in range(count)]
    return (mm, [float(v) for v in ss]) if with_sumsq else mm


ft.minmax_many = timed
sys.argv = ["bench.py", "--config", "5", "--skip-single", "--no-cpu-baseline", "--no-configs"]
bench.main()
for k, v in acc.items():
    print(f"{k:40s} mean {sum(v) / len(v):8.3f} ms over {len(v)} calls (max {max(v):.3f})", file=sys.stderr)
