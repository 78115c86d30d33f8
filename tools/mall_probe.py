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
