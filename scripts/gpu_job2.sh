#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - <<'P' > gpurun_out/r2_pcie.txt 2>&1
import torch, time
for mb in (64, 1024):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(name, mb, "MiB", round(mb / 1024 / dt, 1), "GiB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(1 << 30, dtype=torch.uint8).pin_memory(); d2 = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("both directions at once, 1 GiB each:", round(2 / dt, 1), "GiB/s total")
P
cat gpurun_out/r2_pcie.txt
python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r2_pytest2.log; tail -5 gpurun_out/r2_pytest2.log
MCX_TIMING=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/r2_bench_b.json 2> gpurun_out/r2_bench_b.err
grep "tier 1" gpurun_out/r2_bench_b.err | tail -8
