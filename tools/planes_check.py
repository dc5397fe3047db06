"""How reliably does hgi_planes_alloc separate neighbouring planes?  One fresh process per call (run it in a loop):
prints the verdict, the time the allocation took and the probe times of the three pairings."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
ctx = H.Context(0)
n = 64 * 4096 * 4096
torch.cuda.synchronize()
t0 = time.perf_counter()
p = H.Planes(ctx, n, 3)
dt = time.perf_counter() - t0
a, b, c = p.probe_ms(0, 1), p.probe_ms(1, 2), p.probe_ms(0, 2)
print("separated %-5s  alloc %.0f ms  probe 0->1 %.4f  1->2 %.4f  0->2 %.4f ms   free %.1f GiB" %
      (p.separated, dt * 1e3, a, b, c, torch.cuda.mem_get_info()[0] / 2**30))
p.close()
