#!/usr/bin/env python3
"""Probe: can a file on /dev/shm be mapped, registered with the HIP runtime (hipHostRegister) and copied to the GPU
straight from the page cache, without the pread into a pinned staging buffer records.upload_pixel_major does today?
Prints the H2D rate of (a) pread -> pinned -> device, (b) registered mapping -> device, (c) unregistered mapping -> device."""
import ctypes
import mmap
import os
import sys
import time

import numpy as np
import torch

size = int(float(sys.argv[1]) * 2 ** 30) if len(sys.argv) > 1 else 2 * 2 ** 30
piece = 4096000  # one 80 x 100 x 128 float32 map
path = '/dev/shm/nemo_mmap_probe.bin'
with open(path, 'wb') as f:
    blk = np.random.default_rng(0).integers(0, 255, 64 * 2 ** 20, dtype=np.uint8).tobytes()
    for _ in range(size // len(blk)):
        f.write(blk)
n = size // piece
dst = torch.empty(n * piece, dtype=torch.uint8, device='cuda')
torch.cuda.synchronize()


def rate(fn, label):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{label}: {n * piece / dt / 1e9:.1f} GB/s', flush=True)


# (a) pread into pinned memory with 16 threads, then one copy
from concurrent.futures import ThreadPoolExecutor
pin = torch.empty(n * piece, dtype=torch.uint8).pin_memory()
pin_np = pin.numpy()
fd = os.open(path, os.O_RDONLY)


def read_piece(k):
    os.preadv(fd, [memoryview(pin_np[k * piece:(k + 1) * piece])], k * piece)


def via_pread():
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(read_piece, range(n)))
    dst.copy_(pin, non_blocking=True)


rate(via_pread, 'pread (16 threads) -> pinned -> device')

# (b) / (c) mapping
hip = ctypes.CDLL('libamdhip64.so')
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipGetErrorString.restype = ctypes.c_char_p
for access, aname in ((mmap.ACCESS_COPY, 'private'), (mmap.ACCESS_WRITE, 'shared'), (mmap.ACCESS_READ, 'read-only shared')):
    f = open(path, 'r+b')
    mm = mmap.mmap(f.fileno(), 0, access=access)
    arr = np.frombuffer(mm, dtype=np.uint8)
    if access != mmap.ACCESS_READ:
        src = torch.from_numpy(arr)[:n * piece]
    else:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            src = torch.from_numpy(arr)[:n * piece] if arr.flags.writeable else torch.frombuffer(mm, dtype=torch.uint8)[:n * piece]

    def via_map():
        for k in range(0, n, 64):
            dst[k * piece:(k + 64) * piece].copy_(src[k * piece:(k + 64) * piece], non_blocking=True)

    if access == mmap.ACCESS_COPY:
        rate(via_map, f'{aname} mapping, not registered -> device')
        for T in (1, 2, 4, 8):
            streams = [torch.cuda.Stream() for _ in range(T)]

            def worker(t):
                with torch.cuda.stream(streams[t]):
                    for k in range(t, n, T):
                        dst[k * piece:(k + 1) * piece].copy_(src[k * piece:(k + 1) * piece], non_blocking=True)
                    streams[t].synchronize()

            def via_map_threads():
                with ThreadPoolExecutor(T) as ex:
                    list(ex.map(worker, range(T)))

            rate(via_map_threads, f'{aname} mapping, one 4 MB map per copy, {T} thread(s) / stream(s)')
    ptr = ctypes.addressof(ctypes.c_char.from_buffer(mm)) if access != mmap.ACCESS_READ else src.data_ptr()
    for flags, name in ((0, 'default'), (8, 'read-only flag')):
        rc = hip.hipHostRegister(ptr, arr.nbytes, flags)
        print(f'{aname} mapping: hipHostRegister({name}) -> {rc} {hip.hipGetErrorString(rc).decode()}; is_pinned {src.is_pinned()}', flush=True)
        if rc == 0:
            rate(via_map, f'{aname} mapping, registered ({name}) -> device')
            hip.hipHostUnregister(ptr)
            break
    del src, arr
    try:
        mm.close()
    except BufferError:
        pass
    f.close()
os.close(fd)
os.unlink(path)
