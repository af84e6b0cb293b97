"""Host micro-benchmark: how fast do T threads pread a 1 GB file out of the page cache (8 MB reads into per-thread buffers)?"""
import os, sys, threading, time
path = "/tmp/pread_bw.bin"
N = 1_000_000_000
if not os.path.exists(path) or os.path.getsize(path) != N:
    with open(path, "wb") as f:
        f.write(os.urandom(1 << 20) * (N >> 20) + b"x" * (N - ((N >> 20) << 20)))
with open(path, "rb") as f:
    while f.read(1 << 24):
        pass
CH = 8 << 20
nchunks = (N + CH - 1) // CH
for T in (1, 2, 4, 8, 12, 16, 24, 32):
    fd = os.open(path, os.O_RDONLY)
    nxt = [0]
    lock = threading.Lock()
    def work():
        buf = bytearray(CH)
        while True:
            with lock:
                c = nxt[0]; nxt[0] += 1
            if c >= nchunks: return
            os.preadv(fd, [buf], c * CH)
    best = 1e9
    for _ in range(3):
        nxt[0] = 0
        th = [threading.Thread(target=work) for _ in range(T)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        best = min(best, time.perf_counter() - t)
    os.close(fd)
    print("threads %2d: %.1f GB/s (%.1f ms per GB)" % (T, N / best / 1e9, best * 1e3))
os.remove(path)
