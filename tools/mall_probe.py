"""Does a producer -> consumer pair of streaming kernels run faster when its working set fits the 256 MB memory-side cache
(MI355X "infinity cache")?  Times y = x (torch copy: read x, write y) and then z = y (the consumer reads what the producer
just wrote) for working sets from 32 MB to 2 GB; bytes / time per pair of launches.  Input to the chunked x -> y pipeline
idea of DESIGN section 12."""
import time

import torch


def main():
    dev = torch.device("cuda")
    for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
        n = mb * (1 << 20) // 8
        x = torch.randn(n, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        for _ in range(3):
            y.copy_(x); z.copy_(y)
        torch.cuda.synchronize()
        reps = max(4, 4096 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            y.copy_(x)
            z.copy_(y)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        print("%5d MB per array: %.3f ms per producer+consumer pair, %.2f TB/s (4 x array bytes / time)" %
              (mb, 1e3 * el, 4 * n * 8 / el / 1e12), flush=True)


if __name__ == "__main__":
    main()
