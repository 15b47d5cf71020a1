"""Does fork() from a GPU-initialised process slow that process's GPU work down afterwards?  Launch + synchronize latency of a tiny
kernel, and an H2D copy, before any fork / with 8 forked children alive / after they exited / after 5 more rounds of fork + exit."""
import multiprocessing as mp
import time
import torch


def child(ev):
    ev.wait()


def lat(dev, n=200):
    x = torch.zeros(1024, device=dev)
    h = torch.randn(1, 3, 256, 256)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        x.add_(1.0)
        h.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return f"median {1e6 * ts[n // 2]:.0f} us  p90 {1e6 * ts[int(n * 0.9)]:.0f} us  max {1e3 * ts[-1]:.1f} ms  total {sum(ts):.3f} s"


def main():
    dev = torch.device("cuda", 0)
    big = torch.empty(1 << 28, device=dev)          # 1 GiB of device memory mapped, like a model + workspaces
    pinned = torch.empty(1 << 24).pin_memory()
    print("before any fork      :", lat(dev))
    ctx = mp.get_context("fork")
    for rnd in range(6):
        ev = ctx.Event()
        ps = [ctx.Process(target=child, args=(ev,)) for _ in range(8)]
        for p in ps:
            p.start()
        if rnd == 0:
            print("8 children alive     :", lat(dev))
        ev.set()
        for p in ps:
            p.join()
        if rnd in (0, 5):
            print(f"after {rnd + 1} x (fork 8, exit):", lat(dev))


if __name__ == "__main__":
    main()
