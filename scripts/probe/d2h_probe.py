import torch, ctypes, time
n_out, n_in = 256 << 20, 320 << 20
a = torch.full((n_out,), 3, dtype=torch.uint8, device="cuda")
b = torch.empty(n_in, dtype=torch.uint8, device="cuda")
h = torch.empty(n_out, dtype=torch.uint8).pin_memory()
g = torch.empty(n_in, dtype=torch.uint8).pin_memory()
big = torch.empty(1200 << 20, dtype=torch.uint8, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run(what, d2h, h2d, kern):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if h2d:
            with torch.cuda.stream(s2): b.copy_(g, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s1): h.copy_(a, non_blocking=True)
        if kern:
            with torch.cuda.stream(s3):
                e0.record(); big.add_(1); e1.record()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
    print("%-40s all %.2f ms%s" % (what, t, ", kernel %.2f ms" % e0.elapsed_time(e1) if kern else ""), flush=True)
run("kernel alone", 0, 0, 1)
run("D2H alone", 1, 0, 0)
run("H2D alone", 0, 1, 0)
run("D2H + H2D", 1, 1, 0)
run("D2H + kernel", 1, 0, 1)
run("D2H + H2D + kernel", 1, 1, 1)
