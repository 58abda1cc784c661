import torch, sys
pat = int(sys.argv[1], 0)
ts = []
try:
    for _ in range(60):
        t = torch.empty(1 << 30, dtype=torch.int32, device="cuda:0")   # 4 GiB each
        t.fill_(pat)
        ts.append(t)
except Exception as e:
    pass
torch.cuda.synchronize()
print("polluted", len(ts) * 4, "GiB with", hex(pat))
