import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops
x = torch.randn(4096 * 500, device="cuda")
for name, fn in (("torch.median", lambda: x.median()), ("ops.median", lambda: ops.median(x)),
                 ("double+median", lambda: x.double().median())):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    print(name, "%.1f us" % (s.elapsed_time(e) / 20 * 1e3))
r = torch.rand(4096*500, device="cuda").round()   # two values only: worst case for LDS atomics
for _ in range(3): ops.median(r)
torch.cuda.synchronize(); s.record()
for _ in range(20): ops.median(r)
e.record(); torch.cuda.synchronize(); print("ops.median two-valued", "%.1f us" % (s.elapsed_time(e) / 20 * 1e3))
