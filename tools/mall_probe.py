"""Does a tensor written by one kernel come back from the Infinity Cache when the next kernel reads it?
Times `b.copy_(a)` then `s = b.sum()`-like read passes for tensor sizes around the 256 MB cache."""
import torch, sys
dev = "cuda"
def t(fn, it=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)   # 1 GiB flusher
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb << 20
    a = torch.empty(n // 2, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a); c = torch.empty_like(a)
    # (1) write b then read b (producer -> consumer), interleaved so each pair is back to back
    def pair():
        torch.mul(a, 2.0, out=b)      # reads a, writes b
        torch.add(b, 1.0, out=c)      # reads b (just written), writes c
    def pair_flushed():
        torch.mul(a, 2.0, out=b)
        big.zero_()                   # 1 GiB of writes in between
        torch.add(b, 1.0, out=c)
    def flush_only():
        big.zero_()
    tp, tf, tz = t(pair), t(pair_flushed), t(flush_only)
    print(f"{mb:5d} MB: pair {tp*1e3:8.1f} us ({4*n/tp/1e9:6.2f} TB/s over 4 tensor passes) | with a 1 GiB write between: {(tf-tz)*1e3:8.1f} us ({4*n/(tf-tz)/1e9:6.2f} TB/s)")
