#!/usr/bin/env python3
"""Does device memory come in stretches that do not share their channels?  24 buffers of 1 GiB, allocated one after the other; device-to-device
copies between buffer i and buffer j (torch's copy kernel; ms per GiB copied, i.e. 2 GiB of traffic) for neighbours and for distant pairs, and
plain fills of every buffer.  Usage: python tools/region_probe.py [buffers]"""
import sys


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda", 0)
    gib = 1 << 30
    buffers = [torch.empty(gib, dtype=torch.uint8, device=dev) for _ in range(n)]
    print("addresses (GiB):", " ".join(f"{b.data_ptr() / gib:9.2f}" for b in buffers))

    def timed(fn, repeats=4):
        fn()
        torch.cuda.synchronize()
        started, stopped = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        started.record()
        for _ in range(repeats):
            fn()
        stopped.record()
        torch.cuda.synchronize()
        return started.elapsed_time(stopped) / repeats

    print("fill i              ", " ".join(f"{timed(lambda b=b: b.fill_(1)):6.3f}" for b in buffers))
    print("copy i -> i+1       ", " ".join(f"{timed(lambda i=i: buffers[(i + 1) % n].copy_(buffers[i])):6.3f}" for i in range(n)))
    print("copy i -> i+n/2     ", " ".join(f"{timed(lambda i=i: buffers[(i + n // 2) % n].copy_(buffers[i])):6.3f}" for i in range(n)))
    print("copy 0 -> j         ", " ".join(f"{timed(lambda j=j: buffers[j].copy_(buffers[0])):6.3f}" for j in range(1, n)))
    print("copy n-1 -> j       ", " ".join(f"{timed(lambda j=j: buffers[j].copy_(buffers[n - 1])):6.3f}" for j in range(0, n - 1)))


if __name__ == "__main__":
    main()
