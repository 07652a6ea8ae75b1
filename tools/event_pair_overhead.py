"""Elapsed time of an EMPTY HIP event pair on the stream (what an event bracket adds to a kernel's measured duration)."""
import torch
torch.cuda.init()
x = torch.zeros(1 << 20, device="cuda")
for busy in (False, True):
    tot = 0.0
    K = 200
    for _ in range(K):
        if busy:
            x.add_(1.0)  # a kernel right before the pair, as in the step
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    print("empty pair after %s stream: %.2f us" % ("a busy" if busy else "an idle", tot / K * 1e3))
# a short kernel bracketed vs its rocprof-visible duration is the other half: see DESIGN.md §5
