import torch, time
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    a.fill_(1.0)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    reps = max(5, 4096 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        b.copy_(a)      # write b, read a
        a.copy_(b)      # then read b (just written), write a
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * reps)
    print("copy %5d MB (read + write of the same two buffers, ping-pong): %.2f TB/s" % (mb, 2 * n * 4 / dt / 1e12), flush=True)

# ---- what a consumer finds in the memory-side cache right after a producer wrote 1.5 GB ascending: the newest vs the oldest 192 MB
n = 1536 * 1024 * 1024 // 4
a = torch.empty(n, device="cuda"); src = torch.empty(n, device="cuda"); src.fill_(1.0)
k = 192 * 1024 * 1024 // 4
out = torch.empty(k, device="cuda")
for name, sl in (("oldest", slice(0, k)), ("newest", slice(n - k, n)), ("oldest", slice(0, k)), ("newest", slice(n - k, n))):
    ts = []
    for _ in range(8):
        a.copy_(src)                       # the producer: 1.5 GB written, low addresses first
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.copy_(a[sl])                   # the consumer: 192 MB of it
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("read the %s 192 MB after a 1.5 GB ascending write: %.1f us (median), %.2f TB/s (read + write)" % (name, 1e3 * ts[4], 2 * k * 4 / (ts[4] * 1e-3) / 1e12), flush=True)
