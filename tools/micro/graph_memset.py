"""Does a hipMemsetAsync captured into a HIP graph run on every replay?  (round-2 diagnosis of the graph-replay divergence)
Captures  memset(buf, 0, n) ; buf += 1  and replays it: with working memset nodes buf reads 1 after every replay."""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
for nbytes in (4, 64, 256, 4096):
    n = nbytes // 4
    buf = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        buf += 1
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0, nbytes, C.c_void_p(s))
        buf += 1
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append(buf.tolist()[:2])
    print(nbytes, "rc", rc, "after replays:", vals)
